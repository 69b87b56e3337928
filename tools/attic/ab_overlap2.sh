mkdir -p gpurun_out/r06
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-also --no-cpu-baseline --no-isolated --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '${envs[*]}', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if v>0.05})" >> gpurun_out/r06/ab_overlap2.txt
}
for content in nat g3; do
 for cfg in "48 3" "32 2" "16 2"; do
  set -- $cfg
  A="--tile-w 128 --tile-h 128 --content $content --frames $1 --streams $2"
  for ov in 0 2 1; do run "p128_${content}_$1x$2" LLCOMP_MI_OVERLAP=$ov -- $A; done
 done
done
cat gpurun_out/r06/ab_overlap2.txt
