#!/bin/bash
# Runs on the GPU box: the 2-D tile workload through two builds of the library (LLCOMP_MI_LIB), alternating, one process each.
#   tools/lib_ab_tiles.sh <outdir under gpurun_out> <other library under llcomp_amd/> [contents="g3 nat"] [configs="16x1 48x3"] [reps=2]
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-lib_ab_tiles}; other=$2; contents=${3:-g3 nat}; configs=${4:-16x1 48x3}; reps=${5:-2}
mkdir -p $out
for r in $(seq 1 $reps); do for cfg in $configs; do f=${cfg%x*}; s=${cfg#*x}; for c in $contents; do for lib in libllcomp_mi.so $other; do
  A="--no-cpu-baseline --no-also --frames $f --streams $s --tile-w 64 --tile-h 64 --steps 8 --warmup 2 --content $c"
  LLCOMP_MI_LIB=$PWD/llcomp_amd/$lib timeout -k 10 300 python3 bench.py $A > $out/${lib%.so}_${cfg}_${c}_$r.json 2>/dev/null || exit 1
done; done; done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
    print(f"{os.path.basename(f)[:-5]:44s} {d['value']:8.1f} MPix/s {d['ms_per_step']:8.3f} ms  snapshot pass {k['clear_states_enc']:7.2f}  coder {k['k_encode_slices']:7.2f}  decoder {k['k_decode_slices']:7.2f}")
PY
