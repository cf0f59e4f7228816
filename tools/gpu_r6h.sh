#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -k "split_weights or two_block or conv32" > gpurun_out/r06_xcd_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r06_xcd_tests.log | cut -c1-200
[ $rc -ne 0 ] && exit $rc
for rnd in 1 2; do for o in "conv2b_xcd_walk=1" "conv2b_xcd_walk=0"; do
  for m in bf16 fp16x2; do
    MEDNET_OPTIONS=$o python bench.py --precision $m --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o $m', d['value'], d['ms_per_step'])"
  done
  MEDNET_OPTIONS=$o RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o cfg5 bf16', d['patches_per_s'], d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r06_conv2b_xcd_walk_ab.log
