#!/bin/bash
# Upper bound of what overlapping conv32_mfma_kernel's epilogue with the next brick's tap loop could save:
# a measurement build in which the non-GNB variants skip the epilogue's LDS transposition and the fused statistics altogether
# (the 32 conversions and the 8 row stores stay; the OUTPUT IS WRONG).  Built from a patched COPY of the sources under /tmp, so
# no measurement code lives in csrc/; the library lands next to the product one as libmednet_hip_epibound.so.
# Then: tools/probes/epilogue_bound_ab.sh on the GPU box (profiles/r05_conv32_epilogue_bound.log).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/epib; rm -rf $W; mkdir -p $W; cp $R/torch-mednet_amd/csrc/*.hip $R/torch-mednet_amd/csrc/*.h $R/torch-mednet_amd/csrc/*.inc $W/
python3 - "$R" <<'PY'
import sys
R = sys.argv[1]
c = open('/tmp/epib/common.h').read().replace('#include "../../include/mednet_hip.h"', f'#include "{R}/include/mednet_hip.h"')
open('/tmp/epib/common.h', 'w').write(c)
s = open('/tmp/epib/conv_mfma.hip').read()
a = s.index("    eltx8 rows[8];\n    // LDS operations of a wave execute in order")
b = s.index("    if constexpr (GNB_LDS) {\n      // The rows' LDS-DMAs were this wave")
old = s[a:b]
new = ("    eltx8 rows[8];\n    if constexpr (!GNB) {  // BOUND PROBE: no LDS transposition: the converted accumulators as they lie (WRONG layout)\n"
       "#pragma unroll\n      for (int j = 0; j < 8; ++j)\n#pragma unroll\n        for (int k = 0; k < 8; ++k) rows[j][k] = (elt)acc[j >> 1][(j & 1) * 8 + k];\n"
       "    } else {\n" + old[len("    eltx8 rows[8];\n"):] + "    }\n")
s = s.replace(old, new).replace("      } else if constexpr (STATS) {\n        const eltx8 vz = ok ? v : eltx8{};",
                                "      } else if constexpr (STATS && false) {\n        const eltx8 vz = ok ? v : eltx8{};")
open('/tmp/epib/conv_mfma.hip', 'w').write(s)
PY
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on"
cd $W
/opt/rocm/bin/hipcc $F -c conv_mfma.hip -o conv_mfma.o &
/opt/rocm/bin/hipcc $F -DMEDNET_ELT_F16 -Dmednet=mednet_f16 -c conv_mfma.hip -o conv_mfma_f16.o &
wait
O=$R/torch-mednet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/api.o $O/head_mfma.o $O/head_mfma_f16.o $O/conv_direct.o conv_mfma.o conv_mfma_f16.o $O/conv_f32_mfma.o $O/conv_x3_mfma.o \
  $O/norm_act.o $O/loss.o $O/head_loss.o $O/predict.o $O/augment.o -o $R/torch-mednet_amd/mednet_hip/libmednet_hip_epibound.so
