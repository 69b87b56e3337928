mkdir -p gpurun_out/r06
run() {  # label, env assignments..., --, bench args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-also --no-cpu-baseline --no-isolated --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '${envs[*]}', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if v>0.05})" >> gpurun_out/r06/ab_hwq.txt
}
A="--tile-w 128 --tile-h 128 --content nat --frames 48 --streams 3"
run p128_nat_48x3 LLCOMP_MI_NOSNAP=1 -- $A
run p128_nat_48x3 LLCOMP_MI_NOOVERLAP=1 -- $A
run p128_nat_48x3 LLCOMP_MI_NOOVERLAP=0 -- $A
run p128_nat_48x3 LLCOMP_MI_NOOVERLAP=0 GPU_MAX_HW_QUEUES=8 -- $A
run p128_nat_48x3 LLCOMP_MI_NOOVERLAP=1 GPU_MAX_HW_QUEUES=8 -- $A
run p128_nat_48x3 LLCOMP_MI_NOSNAP=1 GPU_MAX_HW_QUEUES=8 -- $A
A="--tile-w 128 --tile-h 128 --frames 48 --streams 3"
run p128_g3_48x3 LLCOMP_MI_NOOVERLAP=1 -- $A
run p128_g3_48x3 LLCOMP_MI_NOOVERLAP=0 GPU_MAX_HW_QUEUES=8 -- $A
A="--tile-w 128 --tile-h 128 --content nat --frames 16 --streams 1"
run p128_nat_16x1 LLCOMP_MI_NOOVERLAP=1 -- $A
run p128_nat_16x1 LLCOMP_MI_NOOVERLAP=0 GPU_MAX_HW_QUEUES=8 -- $A
cat gpurun_out/r06/ab_hwq.txt
