#!/bin/bash
# round 6: the ConvTranspose3d 64 -> 32 data-gradient specialisation: bit identity with the general kernel, then A/B
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "convt_dgrad32" 2>&1 | tail -15 | tee gpurun_out/r06_convt32_tests.log &&
timeout -k 10 300 python tools/probes/convt_dgrad32_bench.py 2>&1 | tee gpurun_out/r06_convt32_bench.log
