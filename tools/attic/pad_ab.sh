#!/bin/bash
# (round 5, profiles/r05_helper_wait.txt; the second library was a build with another kChunkDwords in model_kernels.hip, or with 16 KB
# of unused LDS added to both kernels: build it from the tree as llcomp_amd/libllcomp_mi_ch32.so before running this)
# Runs on the GPU box: pack / stage with 128-byte chunks (8.4 KB of LDS) against 256-byte chunks (16.6 KB: libllcomp_mi_ch32.so), 64x64 tiles at 48 x 3 and the headline
cd $GRAFT_REPO_ROOT
out=gpurun_out/pad_ab; mkdir -p $out
for r in 1 2; do for lib in libllcomp_mi.so libllcomp_mi_ch32.so; do
  LLCOMP_MI_LIB=$PWD/llcomp_amd/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-also --frames 48 --streams 3 --tile-w 64 --tile-h 64 --steps 8 --warmup 2 --content g3 > $out/${lib%.so}_tiles_$r.json 2>/dev/null || exit 1
  LLCOMP_MI_LIB=$PWD/llcomp_amd/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-also --no-isolated --steps 20 --warmup 3 > $out/${lib%.so}_head_$r.json 2>/dev/null || exit 1
done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
    print(f"{os.path.basename(f)[:-5]:40s} {d['value']:8.1f} {d['ms_per_step']:8.3f}  scan+pack {k['scan+pack']:6.2f} scan+stage {k['k_scan_lengths_dec']:6.2f}")
PY
