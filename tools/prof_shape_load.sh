#!/bin/bash
# Runs on the GPU box: kernel trace + HBM-side counters (separate --pmc passes) of one slicing AT LOAD.
#   tools/prof_shape_load.sh <outdir under gpurun_out> <bench args...>      e.g.  ... r06_i64 --interleaved --tile-w 64 --tile-h 64 --frames 48 --streams 3
set -u
out=gpurun_out/$1; shift
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-isolated --no-also --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $A "$@" > $out/bench_under_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $A "$@" > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $A "$@" > /dev/null 2> $out/w.err
python3 tools/summarize_pmc.py $out > $out/summary.txt
python3 bench.py $A "$@" > $out/bench.json 2> $out/bench.err
