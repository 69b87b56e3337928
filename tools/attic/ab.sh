#!/bin/bash
# GPU box: A/B timing of slice-kernel builds on the headline workload (tools/exp_time.py), alternating, N rounds.
#   tools/ab.sh <outfile under gpurun_out> <rounds> <lib or 'product'> [more libs...]
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift
n=$1; shift
mkdir -p $(dirname $out)
for i in $(seq 1 $n); do
  for lib in "$@"; do
    python3 tools/exp_time.py $lib "round $i" 2>>$out.err | tee -a $out
  done
done
