#!/bin/bash
# A/B of two builds of libmednet_hip.so inside ONE GPU-box call (box-to-box variance is +-5%):
# libmednet_hip_base.so (built from the previous source) vs libmednet_hip.so; kernels micro-benchmark, then the bench.
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
B=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip_base.so
for round in 1 2; do
  echo "== base"; MEDNET_LIB_PATH=$B KB_WHICH=conv python tools/bench_kernels.py 2>&1 | grep "^conv"
  echo "== new";  KB_WHICH=conv python tools/bench_kernels.py 2>&1 | grep "^conv"
done
for round in 1 2; do
  echo "== base"; MEDNET_LIB_PATH=$B python bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-roofline 2>&1 | tail -1 | cut -c1-140
  echo "== new";  python bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-roofline 2>&1 | tail -1 | cut -c1-140
done
