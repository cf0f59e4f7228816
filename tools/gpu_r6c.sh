#!/bin/bash
# round 6, call C: the whole -m gpu suite, then the bench line and config 5 / config 4 with the two-block kernel on and off (same box)
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 900 python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_gpu.log | cut -c1-300
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
for o in "conv2b=1" "conv2b=0" "conv2b=1" "conv2b=0"; do
  echo "== $o"
  MEDNET_OPTIONS=$o python bench.py --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'])"
  MEDNET_OPTIONS=$o RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | tail -1 | cut -c1-160
done 2>&1 | tee gpurun_out/r06_conv2b_step_ab.log
