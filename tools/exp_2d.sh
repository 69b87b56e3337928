#!/bin/bash
# 2-D tile experiments (round 2): random state-bank access microbenchmark + lane-group width sweep.
out=gpurun_out/exp_2d
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
R=tools/ubench/rand_table
{
for alloc in 0 1 2; do for layout in 0 1; do for fl in 0 1 2; do
  timeout -k 5 120 $R 1530 1024 $alloc $layout $fl || echo "FAILED alloc=$alloc layout=$layout fl=$fl"
done; done; done
timeout -k 5 120 $R 6120 512 0 0 0
timeout -k 5 120 $R 6120 512 1 0 0
timeout -k 5 120 $R 380 2048 0 0 0
timeout -k 5 120 $R 380 2048 1 0 0
} > $out/rand_table.txt 2>&1
for alloc in 0 1; do
  rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $out/pmc_rt_$alloc -- $R 1530 1024 $alloc 0 0 > $out/pmc_rt_$alloc.txt 2>&1
done
python3 tools/summarize_pmc.py $out > $out/rand_table_pmc.txt 2>&1
A="--no-cpu-baseline --no-isolated --frames 16 --streams 2 --tile-w 64 --tile-h 64 --steps 3 --warmup 1"
for content in nat mid g3; do for s in 6 5 4 3; do
  echo "== content $content lane_shift $s" >> $out/sweep.txt
  LLCOMP_MI_LANE_SHIFT=$s timeout -k 5 300 python3 bench.py $A --content $content >> $out/sweep.txt 2>> $out/sweep.err
done; done
