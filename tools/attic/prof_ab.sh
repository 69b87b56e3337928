#!/bin/bash
# GPU box: SQ counters of the slice kernels for several builds (tools/prof_slice.py under rocprofv3 --pmc, separate passes).
#   tools/prof_ab.sh <outdir under gpurun_out> <lib> [more libs...]     ('product' = the in-tree library)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/$1; shift
mkdir -p $out
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" != product ]; then export LLCOMP_MI_LIB=$GRAFT_REPO_ROOT/$lib; else unset LLCOMP_MI_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/$tag/a -- python3 tools/prof_slice.py > $out/$tag.a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/$tag/b -- python3 tools/prof_slice.py > $out/$tag.b.log 2>&1
  rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_TRANS SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $out/$tag/c -- python3 tools/prof_slice.py > $out/$tag.c.log 2>&1
  python3 tools/summarize_pmc.py $out/$tag | grep -A12 "k_encode_sl\|k_decode_sl" > $out/$tag.summary.txt
done
