#!/bin/bash
# round 6: config 4 (bf16) with the persistent first layer on / off, interleaved
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
ms() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("ms_per_step"))'; }
for i in 1 2 3; do
  echo "cfg4 bf16: persistent first layer $(RC_WHICH=cfg4only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms   one workgroup per item $(MEDNET_OPTIONS=conv_c1_persist=0 RC_WHICH=cfg4only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms"
  echo "cfg2 bf16: persistent first layer $(python bench.py --warmup 8 --cpu-steps 0 --fp32-steps 0 --no-roofline --steps 30 2>&1 | ms) ms   one workgroup per item $(MEDNET_OPTIONS=conv_c1_persist=0 python bench.py --warmup 8 --cpu-steps 0 --fp32-steps 0 --no-roofline --steps 30 2>&1 | ms) ms"
done 2>&1 | tee gpurun_out/r06_c1_step_ab.log
