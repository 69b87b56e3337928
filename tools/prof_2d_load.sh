#!/bin/bash
# Runs on the GPU box: HBM-side counters of the 2-D tile workload AT LOAD (48 frames, 3 pipelines), separate --pmc passes.
#   tools/prof_2d_load.sh <outdir under gpurun_out> [content=g3]
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-prof_2d_load}
content=${2:-g3}
export TMPDIR=/tmp
mkdir -p $out
A="--no-cpu-baseline --no-isolated --no-also --frames 48 --streams 3 --tile-w 64 --tile-h 64 --steps 4 --warmup 1 --content $content"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $A > $out/bench_under_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $A > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $A > /dev/null 2> $out/w.err
python3 tools/summarize_pmc.py $out > $out/summary.txt
python3 bench.py $A > $out/bench.json 2> $out/bench.err
