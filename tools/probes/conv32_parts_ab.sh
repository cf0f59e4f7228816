#!/bin/bash
# GPU box: conv32_timing.py with the product library and the probe libraries of conv32_parts_build.sh
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
for P in ${LIBS:-"" _c32_nostage _c32_noread _c32_noepi}; do
  echo "== libmednet_hip$P.so"
  MEDNET_LIB_PATH=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip$P.so timeout -k 10 300 python tools/probes/conv32_timing.py 2>&1 | grep "conv32=1"
done | tee gpurun_out/r06_conv32_parts.log
