mkdir -p gpurun_out/r06
run() {  # label, nosnap, bench args...
  local label=$1 ns=$2; shift 2
  LLCOMP_MI_NOSNAP=$ns python bench.py --no-also --no-cpu-baseline --no-isolated --steps 6 --warmup 2 --frames 48 --streams 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label nosnap=$ns', d['value'], d['ms_per_step'], d['config']['compression_ratio'], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/r06/ab_chunked.txt
}
for ns in 1 0 1 0; do run i64_g3 $ns --interleaved --tile-w 64 --tile-h 64; done
for ns in 1 0; do run i64_nat $ns --interleaved --tile-w 64 --tile-h 64 --content nat; done
for ns in 1 0; do run p128_g3 $ns --tile-w 128 --tile-h 128; done
for ns in 1 0; do run p128_nat $ns --tile-w 128 --tile-h 128 --content nat; done
for ns in 1 0; do run p256_nat $ns --tile-w 256 --tile-h 256 --content nat; done
for ns in 1 0; do run p64_g3 $ns --tile-w 64 --tile-h 64; done
cat gpurun_out/r06/ab_chunked.txt
