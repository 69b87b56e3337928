#!/bin/bash
out=gpurun_out/prof_2d
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-isolated --no-also --frames 16 --streams 1 --tile-w 64 --tile-h 64 --steps 3 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $A > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $A > /dev/null 2> $out/w.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- python3 bench.py $A > /dev/null 2> $out/t.err
rocprofv3 --pmc TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum --output-format csv -d $out/pmc_ea -- python3 bench.py $A > /dev/null 2> $out/e.err
python3 tools/summarize_pmc.py $out > $out/summary.txt
