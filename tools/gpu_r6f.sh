#!/bin/bash
# round 6: split-weight mode: its kernel tests, the timed-workload parity, its step rate
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -s -k "split_weights or two_block" > gpurun_out/r06_split_tests.log 2>&1
rc=$?
grep -E "split weights|passed|failed|Error" gpurun_out/r06_split_tests.log | cut -c1-200; tail -5 gpurun_out/r06_split_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_timed_workload.py -m gpu -q -s --tb=short -k "fp16x2 or fp16" > gpurun_out/r06_timed_workload_fp16x2.log 2>&1
rc=$?
grep -E "timed workload|passed|failed|Error" gpurun_out/r06_timed_workload_fp16x2.log | cut -c1-500
for m in fp16x2 fp16 bf16; do
  python bench.py --precision $m --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], d['config']['loss'])"
done
exit $rc
