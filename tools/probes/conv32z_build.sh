#!/bin/bash
# Builds libmednet_hip_conv32z.so: the product library + conv32z_mfma_kernel (tools/probes/conv32z_kernel.inc), option conv32z=1 selects it
# for the plain / statistics variants of the 32 -> 32 forward.  From a patched COPY of the sources under /tmp: no measurement code in csrc/.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/c32z; rm -rf $W; mkdir -p $W; cp $R/torch-mednet_amd/csrc/*.hip $R/torch-mednet_amd/csrc/*.h $R/torch-mednet_amd/csrc/*.inc $W/
python3 - "$R" <<'PY'
import sys
R = sys.argv[1]
c = open('/tmp/c32z/common.h').read().replace('#include "../../include/mednet_hip.h"', f'#include "{R}/include/mednet_hip.h"')
open('/tmp/c32z/common.h', 'w').write(c)
inc = open(f'{R}/tools/probes/conv32z_kernel.inc').read()
kernel = inc[inc.index('//@@KERNEL\n') + len('//@@KERNEL\n'):inc.index('//@@LAUNCHER\n')]
launcher = inc[inc.index('//@@LAUNCHER\n') + len('//@@LAUNCHER\n'):]
s = open('/tmp/c32z/conv_mfma.hip').read()
m1 = '// ================================================================================================== ConvTranspose3d forward'
m2 = '      static bool attr32[16] = {};'
assert m1 in s and m2 in s
s = s.replace(m1, kernel + m1, 1).replace(m2, launcher + m2, 1)
open('/tmp/c32z/conv_mfma.hip', 'w').write(s)
PY
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -mllvm -pragma-unroll-threshold=262144"
cd $W
/opt/rocm/bin/hipcc $F ${C32Z_DEFS:-} -c conv_mfma.hip -o conv_mfma.o &
/opt/rocm/bin/hipcc $F ${C32Z_DEFS:-} -DMEDNET_ELT_F16 -Dmednet=mednet_f16 -c conv_mfma.hip -o conv_mfma_f16.o &
wait
O=$R/torch-mednet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/api.o $O/head_mfma.o $O/head_mfma_f16.o $O/conv_direct.o conv_mfma.o conv_mfma_f16.o $O/conv_f32_mfma.o $O/conv_x3_mfma.o \
  $O/norm_act.o $O/loss.o $O/head_loss.o $O/predict.o $O/augment.o -o $R/torch-mednet_amd/mednet_hip/libmednet_hip_conv32z.so
echo built $R/torch-mednet_amd/mednet_hip/libmednet_hip_conv32z.so
