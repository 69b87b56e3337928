mkdir -p gpurun_out/r06
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-also --no-cpu-baseline --no-isolated --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '${envs[*]}', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if v>0.05})" >> gpurun_out/r06/ab_prio.txt
}
for cfg in "16 1" "32 1" "16 2" "48 3"; do
  set -- $cfg
  A="--tile-w 128 --tile-h 128 --content nat --frames $1 --streams $2"
  for pr in 0 1; do run "p128_nat_$1x$2" LLCOMP_MI_AUX_PRIO=$pr -- $A; done
done
cat gpurun_out/r06/ab_prio.txt
