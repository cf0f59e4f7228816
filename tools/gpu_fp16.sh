#!/bin/bash
# fp16 storage mode at config 2: golden test at 128^3 (prints what it measured) and the bench line with the fp16 sub-record.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 300 python -m pytest tests/test_gpu_network.py -m gpu -q -x -s --tb=short -k "cfg2_128_fp16_storage" > gpurun_out/fp16_test.log 2>&1
rc=$?
grep -E "fp16|passed|failed|Error|assert" gpurun_out/fp16_test.log | tail -12
true

exit $rc
