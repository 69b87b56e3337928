mkdir -p gpurun_out/r06
run() {  # label, nosnap, bench args...
  local label=$1 ns=$2; shift 2
  LLCOMP_MI_NOSNAP=$ns python bench.py --no-also --no-cpu-baseline --no-isolated --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label nosnap=$ns', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if v>0.05})" >> gpurun_out/r06/ab_chunked_load2.txt
}
for cfg in "16 1" "16 2" "32 2" "32 1" "48 3"; do
  set -- $cfg
  for ns in 1 0; do run "p128_nat_f$1_s$2" $ns --tile-w 128 --tile-h 128 --content nat --frames $1 --streams $2; done
  for ns in 1 0; do run "p128_g3_f$1_s$2" $ns --tile-w 128 --tile-h 128 --frames $1 --streams $2; done
  for ns in 1 0; do run "i64_nat_f$1_s$2" $ns --interleaved --tile-w 64 --tile-h 64 --content nat --frames $1 --streams $2; done
done
cat gpurun_out/r06/ab_chunked_load2.txt
