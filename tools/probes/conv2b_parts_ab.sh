#!/bin/bash
# GPU box: conv2b_bench.py (config 2 shapes) with the product library and the three probe libraries of conv2b_parts_build.sh
set -o pipefail
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
for P in ${LIBS:-"" _c2b_noepi _c2b_nobar _c2b_nodma}; do
  echo "== libmednet_hip$P.so"
  MEDNET_LIB_PATH=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip$P.so C2B_WHICH=cfg2 timeout -k 10 300 python tools/probes/conv2b_bench.py 2>&1 | grep "conv " | grep -E " 64->  64| 128-> 128"
done | tee gpurun_out/r06_conv2b_parts.log
