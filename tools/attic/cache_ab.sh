#!/bin/bash
# Runs on the GPU box: the 2-D tile workload (64x64 planar tiles of 4K frames) with the decoder's bank cache in LDS off / 32 / 64
# entries per lane (LLCOMP_MI_CACHE=0|5|6 in the experiment build of commit "2-D decoder: bank cache experiment"; the product knows
# LLCOMP_MI_NOCACHE=1 only: KS="0 5" maps 0 to it), one process each; per-kernel-group times from the library's own events.
#   tools/cache_ab.sh <outdir under gpurun_out> [contents="g3 nat mid"] [configs="16x1 48x3"] [steps=6]
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-cache_ab}
contents=${2:-g3 nat mid}
configs=${3:-16x1 48x3}
steps=${4:-6}
mkdir -p $out
for cfg in $configs; do
  f=${cfg%x*}; s=${cfg#*x}
  for c in $contents; do
    for k in ${KS:-0 5 6}; do
      A="--no-cpu-baseline --no-also --frames $f --streams $s --tile-w 64 --tile-h 64 --steps $steps --warmup 2 --content $c"
      LLCOMP_MI_CACHE=$k LLCOMP_MI_NOCACHE=$([ $k = 0 ] && echo 1 || echo 0) timeout -k 10 300 python3 bench.py $A > $out/c${k}_${cfg}_$c.json 2> $out/c${k}_${cfg}_$c.err || exit 1
    done
  done
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"unreadable",e); continue
    print(os.path.basename(f), d.get("value"), d.get("ms_per_step"), json.dumps(d.get("kernel_ms_per_step")), json.dumps(d.get("isolated_kernel_ms")))
PY
