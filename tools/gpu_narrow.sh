#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_network.py -m gpu -q -x --tb=short -k "narrow or mfma_fwd_dgrad_wgrad or co_resident or groupnorm_backward_sums or cfg5 or any_group_size or conv_act_orders" > gpurun_out/narrow_test.log 2>&1
rc=$?
tail -15 gpurun_out/narrow_test.log
[ $rc -ne 0 ] && exit $rc
for o in "conv_narrow=1" "conv_narrow=0"; do
  echo "== $o"; MEDNET_OPTIONS=$o RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | tail -1 | cut -c1-260
done
MIN_US=60 bash tools/cfg5_trace.sh > /dev/null 2>&1
