#!/bin/bash
# Fused landmark head (head_mfma.hip): its tests, the landmark tests that now run through it, then config 4's rate and trace.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 500 python -m pytest tests/test_gpu_network.py -m gpu -q -x -s --tb=short -k "landmark or cfg4" > gpurun_out/lm_tests.log 2>&1
rc=$?
grep -E "^\[|passed|failed|Error|assert|error" gpurun_out/lm_tests.log | tail -30
[ $rc -eq 0 ] || exit $rc
RC_WHICH=cfg4 RC_PREC=bf16,fp16 timeout -k 10 300 python tools/run_configs.py 2>&1 | tee gpurun_out/lm_cfg4.log | cut -c1-400
timeout -k 10 300 bash tools/cfg4_trace.sh
