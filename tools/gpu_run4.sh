#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -k "mfma or conv" > gpurun_out/pytest_mfma.log 2>&1
echo "mfma rc=$?"; tail -5 gpurun_out/pytest_mfma.log
timeout -k 10 300 python tools/bench_kernels.py > gpurun_out/bench_kernels.log 2>&1; cat gpurun_out/bench_kernels.log
