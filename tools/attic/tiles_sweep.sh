#!/bin/bash
# Runs on the GPU box: 64x64 planar tiles, frames x pipelines sweep, snapshot encoder vs table-in-HBM encoder (LLCOMP_MI_NOSNAP=1).
#   tools/tiles_sweep.sh <outdir under gpurun_out> <content> "<frames:streams> ..."
out=gpurun_out/${1:-tiles_sweep}
c=${2:-g3}
cfgs=${3:-"16:1 16:2 32:2 48:3"}
cd $GRAFT_REPO_ROOT
mkdir -p $out
for cfg in $cfgs; do
  f=${cfg%%:*}; s=${cfg##*:}
  A="--no-cpu-baseline --no-also --no-isolated --frames $f --streams $s --tile-w ${TW:-64} --tile-h ${TH:-64} --steps 4 --warmup 2 --content $c"
  timeout -k 10 300 python3 bench.py $A > $out/snap_${c}_${f}x${s}.json 2> $out/snap_${c}_${f}x${s}.err || exit 1
  LLCOMP_MI_NOSNAP=1 timeout -k 10 300 python3 bench.py $A > $out/nosnap_${c}_${f}x${s}.json 2> $out/nosnap_${c}_${f}x${s}.err || exit 1
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"unreadable",e); continue
    k=d.get("kernel_ms_per_step") or {}
    print(f"{os.path.basename(f):28s} {d.get('value'):>9} MPix/s  {d.get('ms_per_step'):>8} ms/step  enc {k.get('k_model_fwd',0)+k.get('k_encode_slices',0)+k.get('scan+pack',0):7.2f}  dec {k.get('k_scan_lengths_dec',0)+k.get('k_decode_slices',0)+k.get('k_model_inv',0):7.2f}")
PY
