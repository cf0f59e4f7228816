#!/bin/bash
# round 6: lean weight packs (MEDNET_PACK_HIGH_ONLY): the test, then config 5 and config 2 with MEDNET_LEAN_PACK on / off, interleaved
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lean_pack.py -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r06_lean_pack_tests.log || exit 1
grep -q "passed" gpurun_out/r06_lean_pack_tests.log || exit 1
ms() { tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("ms_per_step"))'; }
for i in 1 2 3; do
  echo "cfg5 bf16: lean packs $(RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms   full packs $(MEDNET_LEAN_PACK=0 RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | ms) ms"
  echo "cfg2 bf16: lean packs $(python bench.py --warmup 8 --cpu-steps 0 --fp32-steps 0 --no-roofline --steps 30 2>&1 | ms) ms   full packs $(MEDNET_LEAN_PACK=0 python bench.py --warmup 8 --cpu-steps 0 --fp32-steps 0 --no-roofline --steps 30 2>&1 | ms) ms"
done 2>&1 | tee gpurun_out/r06_lean_pack_step_ab.log
