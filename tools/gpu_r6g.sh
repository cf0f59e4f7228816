#!/bin/bash
# round 6: race screen of conv2b, cfg5 golden in fp16x2, config-5 / config-4 rates, conv2b fill threshold A/B on config 5
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 600 python tools/probes/conv2b_soak.py > gpurun_out/r06_conv2b_race_screen.log 2>&1; echo "soak rc=$?"; tail -14 gpurun_out/r06_conv2b_race_screen.log
timeout -k 10 900 python -m pytest tests/test_gpu_network.py -m gpu -q -s --tb=short -k "cfg5_full_size_against_reference_golden" > gpurun_out/r06_cfg5_golden.log 2>&1; echo "cfg5 golden rc=$?"
grep -a "cfg5 160x160x96\|passed\|failed" gpurun_out/r06_cfg5_golden.log | cut -c1-300
for o in "conv2b_min_fill=80" "conv2b_min_fill=70" "conv2b_min_fill=80" "conv2b_min_fill=70"; do
  echo "== $o"; MEDNET_OPTIONS=$o RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | tail -1 | cut -c1-170
done 2>&1 | tee gpurun_out/r06_cfg5_fill_ab.log
