#!/bin/bash
# One GPU-box call for the co-resident weight gradient: its parity test, then the stand-alone / side-by-side timings.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -k "co_resident_weight_gradient" > gpurun_out/wg4_test.log 2>&1
rc=$?
tail -15 gpurun_out/wg4_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/probes/wgrad4_bench.py 2>&1 | tee gpurun_out/wg4_bench.log
