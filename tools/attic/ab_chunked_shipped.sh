# the shipped chunked snapshot pass (second stream per device, high priority) against the table encoder, over slicing x content x load
mkdir -p gpurun_out/r06
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-also --no-cpu-baseline --no-isolated --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '${envs[*]}', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if v>0.05})" >> gpurun_out/r06/ab_chunked_shipped.txt
}
for shape in "p128 --tile-w 128 --tile-h 128" "i64 --interleaved --tile-w 64 --tile-h 64"; do
 set -- $shape; name=$1; shift; S="$@"
 for content in nat g3; do
  for cfg in "16 1" "16 2" "32 2" "48 3"; do
   set -- $cfg
   for ns in 1 0; do run "${name}_${content}_$1x$2" LLCOMP_MI_NOSNAP=$ns -- $S --content $content --frames $1 --streams $2; done
  done
 done
done
cat gpurun_out/r06/ab_chunked_shipped.txt
