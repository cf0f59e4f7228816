#!/bin/bash
# Round 6: where does conv32_mfma_kernel's time go?  Throwaway libraries from patched COPIES of the sources (outputs WRONG):
#   libmednet_hip_c32_nostage.so -- no global loads and no LDS commits of the next bricks inside the tap loop (stale bricks)
#   libmednet_hip_c32_noread.so  -- the six operand fragments of a group are read once per brick only (no LDS reads in the tap loop)
#   libmednet_hip_c32_noepi.so   -- no LDS transposition / statistics in the epilogue (raw accumulators stored)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/torch-mednet_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -mllvm -pragma-unroll-threshold=262144"
for P in ${PARTS:-nostage noread noepi}; do
  W=/tmp/c32_$P; rm -rf $W; mkdir -p $W; cp $O/*.hip $O/*.h $O/*.inc $W/
  python3 - "$R" "$W" "$P" <<'PY'
import sys
R, W, P = sys.argv[1:4]
c = open(W + '/common.h').read().replace('#include "../../include/mednet_hip.h"', f'#include "{R}/include/mednet_hip.h"')
open(W + '/common.h', 'w').write(c)
s = open(W + '/conv_mfma.hip').read()
if P == "nostage":
    old = "        if (staging) {\n          if (t == 0) commit_one(buf ^ 1, round);"
    assert old in s
    s = s.replace(old, "        if (staging && false) {\n          if (t == 0) commit_one(buf ^ 1, round);")
elif P == "noread":
    old = "        if (gq + 1 < 18 && (t & 1) == 0) xf[cur ^ 1][(ky * 4 + t) >> 1] = b_fragment(gq + 1, (ky * 4 + t) >> 1);"
    assert old in s
    s = s.replace(old, "        if (gq + 1 < 18 && (t & 1) == 0 && gq == 0) xf[cur ^ 1][(ky * 4 + t) >> 1] = b_fragment(gq + 1, (ky * 4 + t) >> 1);")
elif P == "noepi":
    a = s.index("    eltx8 rows[8];\n    // LDS operations of a wave execute in order")
    b = s.index("    if constexpr (GNB_LDS) {\n      // The rows' LDS-DMAs were this wave")
    s = s[:a] + ("    eltx8 rows[8];\n#pragma unroll\n    for (int j = 0; j < 8; ++j)\n#pragma unroll\n      for (int k = 0; k < 8; ++k) rows[j][k] = (elt)acc[j >> 1][(j & 1) * 8 + k];\n") + s[b:]
    s = s.replace("      } else if constexpr (STATS) {\n        const eltx8 vz = ok ? v : eltx8{};", "      } else if constexpr (STATS && false) {\n        const eltx8 vz = ok ? v : eltx8{};")
open(W + '/conv_mfma.hip', 'w').write(s)
PY
  ( cd $W && /opt/rocm/bin/hipcc $F -c conv_mfma.hip -o conv_mfma.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/api.o $O/head_mfma.o $O/head_mfma_f16.o $O/conv_direct.o \
      conv_mfma.o $O/conv_mfma_f16.o $O/conv_f32_mfma.o $O/conv_x3_mfma.o $O/norm_act.o $O/loss.o $O/head_loss.o $O/predict.o $O/augment.o \
      -o $R/torch-mednet_amd/mednet_hip/libmednet_hip_c32_$P.so && echo built $P ) &
done
wait
