#!/bin/bash
# What does the general convolution kernel lose when ONE workgroup per CU runs instead of two (its epilogues and chunk barriers are
# then nobody's to hide)?  The price a two-channel-blocks-per-wave form would pay for its 128 accumulator registers at one wave per
# SIMD (profiles/r05_ab.md section 10).  Measurement build from a patched COPY of the sources under /tmp: launch bounds (256, 1) and a
# persistent grid of 256; results stay right.  -> libmednet_hip_1wg.so next to the product library; then tools/probes/one_wg_per_cu_ab.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/onewg; rm -rf $W; mkdir -p $W; cp $R/torch-mednet_amd/csrc/*.hip $R/torch-mednet_amd/csrc/*.h $R/torch-mednet_amd/csrc/*.inc $W/
python3 - "$R" <<'PY'
import sys
R = sys.argv[1]
c = open('/tmp/onewg/common.h').read().replace('#include "../../include/mednet_hip.h"', f'#include "{R}/include/mednet_hip.h"')
open('/tmp/onewg/common.h', 'w').write(c)
s = open('/tmp/onewg/conv_mfma.hip').read()
a = "__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(FwdArgs a) {"
assert s.count(a) == 1
s = s.replace(a, "__global__ __launch_bounds__(256, 1) void conv_mfma_kernel(FwdArgs a) {")
a = "if (tuning_option(\"conv_persist\", 1) && grid > 512u) grid = 512u;"
assert s.count(a) == 1
s = s.replace(a, "if (tuning_option(\"conv_persist\", 1) && grid > 256u) grid = 256u;")
open('/tmp/onewg/conv_mfma.hip', 'w').write(s)
PY
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -mllvm -pragma-unroll-threshold=262144"
cd $W
/opt/rocm/bin/hipcc $F -c conv_mfma.hip -o conv_mfma.o &
/opt/rocm/bin/hipcc $F -DMEDNET_ELT_F16 -Dmednet=mednet_f16 -c conv_mfma.hip -o conv_mfma_f16.o &
wait
O=$R/torch-mednet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/api.o $O/conv_direct.o conv_mfma.o conv_mfma_f16.o $O/conv_f32_mfma.o $O/conv_x3_mfma.o \
  $O/norm_act.o $O/loss.o $O/head_loss.o $O/head_mfma.o $O/head_mfma_f16.o $O/predict.o $O/augment.o -o $R/torch-mednet_amd/mednet_hip/libmednet_hip_1wg.so
echo built
