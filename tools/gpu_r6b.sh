#!/bin/bash
# round 6, call B: the two-block kernel's bit-identity test, then its micro-benchmark against the general kernel
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -k "two_block" > gpurun_out/r06_conv2b_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r06_conv2b_tests.log | cut -c1-300
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/probes/conv2b_bench.py > gpurun_out/r06_conv2b_bench.log 2>&1 || { tail -20 gpurun_out/r06_conv2b_bench.log; exit 1; }
cat gpurun_out/r06_conv2b_bench.log
