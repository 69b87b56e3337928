#!/bin/bash
# Runs on the GPU box: the headline workload with slice offsets from the chained scan inside pack / stage (default) and from the
# scan kernels in front of them (LLCOMP_MI_SCANKERNELS=1), alternating, one process each.
#   tools/scan_ab.sh <outdir under gpurun_out> [repeats=3] ["extra bench args"]
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-scan_ab}
reps=${2:-3}
extra=$3
mkdir -p $out
for r in $(seq 1 $reps); do
  for k in 0 1; do
    LLCOMP_MI_SCANKERNELS=$k timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 $extra > $out/k${k}_$r.json 2> $out/k${k}_$r.err || exit 1
  done
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["kernel_ms_per_step"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "scan+pack", k["scan+pack"], "scan_dec", k["k_scan_lengths_dec"], "enc", k["k_encode_slices"], "dec", k["k_decode_slices"], "fwd", k["k_model_fwd"], "inv", k["k_model_inv"])
PY
