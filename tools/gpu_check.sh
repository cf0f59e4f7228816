#!/bin/bash
# One GPU-box call: build check, smoke, the -m gpu suite (all failures reported, not just the first).
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
( rocminfo | grep -m1 -E "gfx9" ; nproc ; free -g | head -2 ) > gpurun_out/box.txt 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -q -x --tb=short "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -40 gpurun_out/pytest_gpu.log
echo "pytest rc=$rc"
exit $rc
