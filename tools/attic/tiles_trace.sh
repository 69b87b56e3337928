#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the 2-D tile workload (16 frames 4K, 64x64 planar tiles, one pipeline).
#   tools/tiles_trace.sh <outdir under gpurun_out> [content=g3] [extra bench args]
out=gpurun_out/${1:-tiles_trace}
content=${2:-g3}
shift $(( $# < 2 ? $# : 2 ))
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p $out
A="--no-cpu-baseline --no-isolated --no-also --frames 16 --streams 1 --tile-w 64 --tile-h 64 --steps 3 --warmup 1 --content $content $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $A > $out/bench_under_trace.json 2> $out/trace.err || exit 1
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
cp $f $out/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
