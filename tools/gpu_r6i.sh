#!/bin/bash
# round 6 against round 5's library (libmednet_hip_base.so built from commit a839d20's csrc) on ONE box, interleaved: configs 2, 4, 5
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
BASE=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip_base.so
ms() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("ms_per_step"))'; }
B="python bench.py --warmup 8 --cpu-steps 0 --fp32-steps 0 --no-roofline --steps 30"
for i in 1 2 3; do
  echo "cfg2 bf16: round 6 $($B 2>&1 | ms) ms   round 5 $(MEDNET_LIB_PATH=$BASE $B 2>&1 | ms) ms"
  echo "cfg4 bf16: round 6 $(RC_WHICH=cfg4only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms   round 5 $(MEDNET_LIB_PATH=$BASE RC_WHICH=cfg4only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms"
  echo "cfg5 bf16: round 6 $(RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms   round 5 $(MEDNET_LIB_PATH=$BASE RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms"
done 2>&1 | tee gpurun_out/r06_vs_r05_same_box.log
