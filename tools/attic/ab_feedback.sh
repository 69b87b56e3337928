mkdir -p gpurun_out/r06
A="--content mid --tile-w 64 --tile-h 64 --frames 48 --streams 3 --no-also --no-cpu-baseline --no-isolated --steps 12 --warmup 3"
for i in 1 2; do
 for fb in 0 1; do
  LLCOMP_MI_NOFEEDBACK=$fb python bench.py $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nofeedback=$fb', d['value'], d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/r06/ab_feedback.txt
 done
done
cat gpurun_out/r06/ab_feedback.txt
