#!/bin/bash
# usage: tools/sweep.sh out.log  "args1" "args2" ...   (runs on the GPU box)
out=$1; shift
: > $out
for a in "$@"; do python3 bench.py --no-cpu-baseline $a >> $out 2>&1; done
