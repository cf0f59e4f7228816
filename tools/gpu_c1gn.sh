#!/bin/bash
# Fused first-layer backward (GroupNorm apply inside the Cin = 1 weight gradient): parity tests, then the step with / without it.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 300 python -m pytest tests/test_gpu_network.py -m gpu -q -x --tb=short -k "first_layer_groupnorm or groupnorm_backward_sums" > gpurun_out/c1gn_tests.log 2>&1
rc=$?
tail -15 gpurun_out/c1gn_tests.log
[ $rc -eq 0 ] || exit $rc
AB="fused:MEDNET_FUSE_C1GN=1;two_kernels:MEDNET_FUSE_C1GN=0" timeout -k 10 400 bash tools/ab_options.sh 2>&1 | tee gpurun_out/c1gn_ab.log
