#!/bin/bash
# Runs on the GPU box: the profiles the judge reads, for the DEFAULT bench command.
#   tools/prof_round.sh <outdir under gpurun_out> [bench args...]
set -u
out=gpurun_out/$1; shift
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-isolated --no-also "$@" > $out/bench_under_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-isolated --no-also "$@" > $out/bench_under_pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --no-cpu-baseline --no-isolated --no-also "$@" > $out/bench_under_pmc_write.json 2> $out/pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- python3 bench.py --no-cpu-baseline --no-isolated --no-also "$@" > $out/bench_under_pmc_tcc.json 2> $out/pmc_tcc.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_sq1 -- python3 bench.py --no-cpu-baseline --no-isolated --no-also "$@" > /dev/null 2> $out/pmc_sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq2 -- python3 bench.py --no-cpu-baseline --no-isolated --no-also "$@" > /dev/null 2> $out/pmc_sq2.err
python3 tools/summarize_pmc.py $out > $out/summary.txt
python3 bench.py --no-also "$@" > $out/bench.json 2> $out/bench.err
