#!/bin/bash
# Full -m gpu suite, then config 4's rate and trace.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 900 python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -8 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
RC_WHICH=cfg4 RC_PREC=bf16 timeout -k 10 300 python tools/run_configs.py 2>&1 | tee gpurun_out/lm_cfg4.log | cut -c1-300
timeout -k 10 300 bash tools/cfg4_trace.sh | grep -E "head_lm|step wall|wgrad_c1"
