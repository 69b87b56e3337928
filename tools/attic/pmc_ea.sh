cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_ea; mkdir -p $out
A="--no-cpu-baseline --no-isolated --no-also --frames 16 --streams 1 --tile-w 64 --tile-h 64 --steps 3 --warmup 1 --content g3"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/rd -- python3 bench.py $A > /dev/null 2> $out/rd.err
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/wr -- python3 bench.py $A > /dev/null 2> $out/wr.err
python3 tools/summarize_pmc.py $out > $out/summary.txt
grep -A4 "k_decode_sl\|k_snap_unperm\|k_model_f" $out/summary.txt | head -60
