#!/bin/bash
# round 6: the persistent first-layer kernel: tests that touch it, then its stand-alone A/B
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "first_layer or split_weights or fused_gn or gcr" 2>&1 | tail -15 | tee gpurun_out/r06_c1_tests.log &&
timeout -k 10 300 python tools/probes/c1_fwd_bench.py 2>&1 | tee gpurun_out/r06_c1_bench.log
