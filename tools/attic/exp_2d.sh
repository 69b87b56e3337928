#!/bin/bash
# 2-D tile experiments (round 2): random state-bank access microbenchmark -> profiles/r02_rand_table.txt.
# Build first: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/rand_table tools/ubench/rand_table.hip
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
# (rocprofv3 --pmc aborts on this bare HIP executable on the round-2 image; the slice kernels' own counters are in
#  profiles/r01_tiles64_f16_pmc_summary.txt.)  Shape and lane-group sweeps: tools/shape_sweep.py.
