mkdir -p gpurun_out/r06
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-also --no-cpu-baseline --no-isolated --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '${envs[*]}', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if v>0.05})" >> gpurun_out/r06/ab_chunked_final.txt
}
for content in nat g3; do
 for cfg in "48 3" "16 1" "32 1"; do
  set -- $cfg
  A="--interleaved --tile-w 64 --tile-h 64 --content $content --frames $1 --streams $2"
  for ns in 1 0; do run "i64_${content}_$1x$2" LLCOMP_MI_NOSNAP=$ns -- $A; done
 done
done
for content in nat g3; do
 for cfg in "16 1" "32 1"; do
  set -- $cfg
  A="--tile-w 128 --tile-h 128 --content $content --frames $1 --streams $2"
  for ns in 1 0; do run "p128_${content}_$1x$2" LLCOMP_MI_NOSNAP=$ns -- $A; done
 done
done
cat gpurun_out/r06/ab_chunked_final.txt
