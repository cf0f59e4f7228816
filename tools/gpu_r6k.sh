#!/bin/bash
# round 6: the ConvTranspose3d data-gradient specialisation in the step: same box, same library, option on / off, configs 2 and 4
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
ms() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("ms_per_step"))'; }
B="python bench.py --warmup 8 --cpu-steps 0 --fp32-steps 0 --no-roofline --steps 30"
for i in 1 2 3; do
  echo "cfg2 bf16: convt_dgrad32 on $($B 2>&1 | ms) ms   off $(MEDNET_OPTIONS=convt_dgrad32=0 $B 2>&1 | ms) ms"
  echo "cfg4 bf16: convt_dgrad32 on $(RC_WHICH=cfg4only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms   off $(MEDNET_OPTIONS=convt_dgrad32=0 RC_WHICH=cfg4only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms"
done 2>&1 | tee gpurun_out/r06_convt32_step_ab.log
