#!/bin/bash
# A/B of the whole training step inside ONE GPU-box call: libmednet_hip_base.so vs libmednet_hip.so (+ option strings)
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
B=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip_base.so
run() { python bench.py --steps 10 --warmup 3 --cpu-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'patches/s', d['ms_per_step'], 'ms', 'dominant', d['roofline']['avg_ms'], 'ms', d['roofline']['frac'])"; }
for round in 1 2 3; do
  MEDNET_LIB_PATH=$B run base
  MEDNET_OPTIONS=${AB_OPT0:-conv_persist=0} run "new ${AB_OPT0:-conv_persist=0}"
  MEDNET_OPTIONS=${AB_OPT1:-conv_persist=1} run "new ${AB_OPT1:-conv_persist=1}"
done
