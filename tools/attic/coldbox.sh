#!/bin/bash
# diagnostic: does the config-4 leg run slower in the first process on a freshly provisioned box?
cd $GRAFT_REPO_ROOT
(rocm-smi --showclocks --showperflevel --showmeminfo vram 2>&1 | grep -v "^=\|^$" | head -20) > gpurun_out/cold_smi0.txt
for i in 1 2 3; do
  python bench.py --also-only c4 --no-cpu-baseline --no-isolated --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i headline', d['value'], 'c4 ms', d['also']['c4_sharded_one_gpu']['ms_per_step'])" | tee -a gpurun_out/cold_runs.txt
done
(rocm-smi --showclocks --showperflevel 2>&1 | grep -v "^=\|^$" | head -20) > gpurun_out/cold_smi1.txt
