#!/bin/bash
# GPU box: headline value over pipelines x frames (bench.py --no-also --no-cpu-baseline --no-isolated)
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift
for cfg in "$@"; do
  s=${cfg%%x*}; f=${cfg##*x}
  python3 bench.py --no-also --no-cpu-baseline --no-isolated --streams $s --frames $f --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s frames $f', d['value'], d['ms_per_step'])" | tee -a $out
done
