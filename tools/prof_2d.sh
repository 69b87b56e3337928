#!/bin/bash
# Runs on the GPU box: HBM-side counters of the 2-D tile kernels (state tables in HBM), separate --pmc passes.
#   tools/prof_2d.sh <outdir under gpurun_out> [content=g3]
out=gpurun_out/${1:-prof_2d}
content=${2:-g3}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p $out
A="--no-cpu-baseline --no-isolated --no-also --frames 16 --streams 1 --tile-w 64 --tile-h 64 --steps 3 --warmup 1 --content $content"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $A > $out/bench_under_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $A > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $A > /dev/null 2> $out/w.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- python3 bench.py $A > /dev/null 2> $out/t.err
python3 tools/summarize_pmc.py $out > $out/summary.txt
python3 bench.py $A > $out/bench.json 2> $out/bench.err
