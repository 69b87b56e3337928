#!/bin/bash
# Runs on the GPU box: instruction-issue counters of the slice kernels (what the VALU/SALU ports and the fetcher do).
# usage: tools/prof_issue.sh <outdir under gpurun_out> [bench args...]
set -u
out=gpurun_out/$1; shift
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_a -- python3 bench.py --no-cpu-baseline --no-also "$@" > /dev/null 2> $out/pmc_a.err
rocprofv3 --pmc SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VSKIPPED SQ_IFETCH_LEVEL --output-format csv -d $out/pmc_b -- python3 bench.py --no-cpu-baseline --no-also "$@" > /dev/null 2> $out/pmc_b.err
python3 tools/summarize_pmc.py $out > $out/summary.txt
