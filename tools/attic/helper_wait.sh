#!/bin/bash
# Runs on the GPU box: kernel trace of the default workload through two builds of the library: how long do the helper kernels take
# (average / maximum) beside the slice kernels?   tools/helper_wait.sh <outdir under gpurun_out> <other library under llcomp_amd/>
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-helper_wait}; other=$2
export TMPDIR=/tmp
mkdir -p $out
for lib in libllcomp_mi.so $other; do
  LLCOMP_MI_LIB=$PWD/llcomp_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out/${lib%.so} -- python3 bench.py --no-cpu-baseline --no-isolated --no-also --steps 20 --warmup 3 > $out/${lib%.so}.json 2> $out/${lib%.so}.err
done
for r in 1 2 3; do for lib in libllcomp_mi.so $other; do
  LLCOMP_MI_LIB=$PWD/llcomp_amd/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-isolated --no-also --steps 20 --warmup 3 > $out/${lib%.so}_run$r.json 2>/dev/null
done; done
python3 - <<PY
import csv,glob,json,os
for d in sorted(glob.glob("$out/libllcomp_mi*")):
    if not os.path.isdir(d): continue
    f=glob.glob(d+"/**/*kernel_stats.csv",recursive=True)[0]
    print(os.path.basename(d))
    for r in csv.DictReader(open(f)):
        if "llcomp_mi" in r["Name"]:
            import re
            k=re.search(r"(k_[a-z_0-9]+)",r["Name"]).group(1)
            print(f"   {k:22s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
for f in sorted(glob.glob("$out/*_run*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d["value"], d["ms_per_step"])
PY
