#!/bin/bash
# Builds and runs the hipFree / hipMalloc churn reproducer (free_wipe.hip) against both HIP runtimes a process of this
# project can hold: /opt/rocm's (what the C++ tools link) and the one PyTorch bundles (what the Python tests load).
# usage: tools/ubench/run_free_wipe.sh [iterations] > gpurun_out/free_wipe.txt
set -e
cd "$(dirname "$0")"
IT=${1:-300}
hipcc -O2 --offload-arch=gfx950 -o free_wipe free_wipe.hip
echo "== /opt/rocm runtime, churn with hipFree =="
timeout -k 10 300 ./free_wipe "$IT" 0
echo "== /opt/rocm runtime, control (never free) =="
timeout -k 10 300 ./free_wipe "$IT" 1
TORCH_LIB=$(python3 -c 'import importlib.util,os;print(os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin),"lib"))')
if [ -e "$TORCH_LIB/libamdhip64.so" ]; then
    hipcc -O2 --offload-arch=gfx950 -no-hip-rt -o free_wipe_torchrt free_wipe.hip -L"$TORCH_LIB" -lamdhip64 -Wl,-rpath,"$TORCH_LIB"
    echo "== PyTorch's bundled HIP runtime ($TORCH_LIB), churn with hipFree =="
    timeout -k 10 300 ./free_wipe_torchrt "$IT" 0
fi
