#!/bin/bash
# Runs on the GPU box: the same bench.py workload through two builds of the library (LLCOMP_MI_LIB), one process each, repeated.
#   tools/lib_ab.sh <outdir under gpurun_out> <other library> "<bench args>" [repeats=2]
out=gpurun_out/${1:-lib_ab}
other=$2
args=$3
reps=${4:-2}
cd $GRAFT_REPO_ROOT
mkdir -p $out
for r in $(seq 1 $reps); do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-also --no-isolated $args > $out/base_$r.json 2> $out/base_$r.err || exit 1
  LLCOMP_MI_LIB=$PWD/$other timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-also --no-isolated $args > $out/other_$r.json 2> $out/other_$r.err || exit 1
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d.get("kernel_ms_per_step") or {}
    print(f"{os.path.basename(f):16s} {d.get('value'):>9} MPix/s {d.get('ms_per_step'):>8} ms/step  " + " ".join(f"{a}={b}" for a,b in k.items() if b>0.3))
PY
