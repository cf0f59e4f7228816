#!/bin/bash
# Two channel blocks per work item in the general convolution kernel: bit-identity tests, layer timings, step A/B.  Needs the library
# built from the patched sources (git apply tools/probes/conv_nb2.patch; make): the option conv_nb2 exists only there.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -k "two_channel_blocks or narrow_bricks" > gpurun_out/nb2_tests.log 2>&1
rc=$?
tail -6 gpurun_out/nb2_tests.log
[ $rc -eq 0 ] || exit $rc
for o in 1 0 1 0; do echo "== conv_nb2=$o"; MEDNET_OPTIONS=conv_nb2=$o python tools/probes/one_wg_per_cu_ab.py 2>&1 | grep "@"; done | tee gpurun_out/nb2_layers.log
AB="nb2:MEDNET_OPTIONS=conv_nb2=1;nb1:MEDNET_OPTIONS=conv_nb2=0" timeout -k 10 400 bash tools/ab_options.sh 2>&1 | tee gpurun_out/nb2_ab.log
for o in 1 0 1 0; do echo "== cfg5 conv_nb2=$o"; MEDNET_OPTIONS=conv_nb2=$o RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | tail -1 | cut -c1-200; done | tee gpurun_out/nb2_cfg5.log
