#!/bin/bash
# Runs on the GPU box: the 2-D tile workload (16 frames 4K, 64x64 planar tiles, one pipeline) with the snapshot encoder and with
# the round-3 table-in-HBM encoder (LLCOMP_MI_NOSNAP=1), one process each; per-kernel-group times from the library's own events.
#   tools/tiles_ab.sh <outdir under gpurun_out> [contents="g3 nat mid"] [streams=1]
out=gpurun_out/${1:-tiles_ab}
contents=${2:-g3 nat mid}
streams=${3:-1}
cd $GRAFT_REPO_ROOT
mkdir -p $out
for c in $contents; do
  A="--no-cpu-baseline --no-also --frames 16 --streams $streams --tile-w 64 --tile-h 64 --steps 5 --warmup 2 --content $c"
  timeout -k 10 300 python3 bench.py $A > $out/snap_$c.json 2> $out/snap_$c.err || exit 1
  LLCOMP_MI_NOSNAP=1 timeout -k 10 300 python3 bench.py $A > $out/nosnap_$c.json 2> $out/nosnap_$c.err || exit 1
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
