#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + SQ counter passes for one bench configuration.
# usage: tools/prof_counters.sh <tag> <bench args...>
set -u
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc1 -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_pmc1.json 2> $out/pmc1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $out/pmc2 -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_pmc2.json 2> $out/pmc2.err
find $out -name "*.csv" | head -20
