#!/bin/bash
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
B=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip_base.so
T=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip_timing.so
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mfma" 2>&1 | tail -2 || exit 1
echo "== timing persist=1"; MEDNET_LIB_PATH=$T timeout -k 10 300 python tools/probes/conv_timing.py 2>&1 | grep -v amdgpu
echo "== timing persist=0"; MEDNET_OPTIONS=conv_persist=0 MEDNET_LIB_PATH=$T timeout -k 10 300 python tools/probes/conv_timing.py 2>&1 | grep -v amdgpu | head -16
for round in 1 2; do
  echo "== new, conv_persist=0"; MEDNET_OPTIONS=conv_persist=0 KB_WHICH=conv python tools/bench_kernels.py 2>&1 | grep "conv \|fwd +"
  echo "== new, conv_persist=1"; KB_WHICH=conv python tools/bench_kernels.py 2>&1 | grep "conv \|fwd +"
done
