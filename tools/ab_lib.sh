#!/bin/bash
# Same-box A/B of two builds of the library: the tree's libmednet_hip.so against another build (default: a copy named
# libmednet_hip_base.so, e.g. built from the previous commit), bf16 and fp32 storage, alternating.  Run on the GPU box.
BASE=${1:-$PWD/torch-mednet_amd/mednet_hip/libmednet_hip_base.so}
ms() { tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])'; }
B="python bench.py --warmup 8 --cpu-steps 0 --fp32-steps 0 --no-roofline"
for i in 1 2 3; do
  echo "bf16 new  $($B --steps 30 2>&1 | ms)   base $(MEDNET_LIB_PATH=$BASE $B --steps 30 2>&1 | ms)"
  echo "fp32 new  $($B --precision fp32 --steps 12 2>&1 | ms)   base $(MEDNET_LIB_PATH=$BASE $B --precision fp32 --steps 12 2>&1 | ms)"
done
