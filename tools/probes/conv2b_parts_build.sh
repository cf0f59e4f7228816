#!/bin/bash
# Round 6: where does conv2b_mfma_kernel's time go?  Three throwaway libraries built from patched COPIES of the sources under /tmp
# (no measurement code in csrc/; the OUTPUTS ARE WRONG):
#   libmednet_hip_c2b_noepi.so   -- the epilogue's LDS transposition, statistics and second operands skipped (stores of the raw
#                                   accumulator registers stay): what hiding the epilogue behind the next item could save at most
#   libmednet_hip_c2b_nobar.so   -- no s_barrier in the phases (waves run free): what the three barriers per chunk cost
#   libmednet_hip_c2b_nodma.so   -- no LDS-DMA inside the tap loops (stale operands): what issuing / landing them costs
# Then: tools/probes/conv2b_bench.py with MEDNET_LIB_PATH set to each (tools/probes/conv2b_parts_ab.sh on the GPU box).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/torch-mednet_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -mllvm -pragma-unroll-threshold=262144"
for P in ${PARTS:-noepi nobar nodma}; do
  W=/tmp/c2b_$P; rm -rf $W; mkdir -p $W; cp $O/*.hip $O/*.h $O/*.inc $W/
  python3 - "$R" "$W" "$P" <<'PY'
import sys
R, W, P = sys.argv[1:4]
c = open(W + '/common.h').read().replace('#include "../../include/mednet_hip.h"', f'#include "{R}/include/mednet_hip.h"')
open(W + '/common.h', 'w').write(c)
s = open(W + '/conv2b_mfma.inc').read()
if P == "noepi":
    a = s.index("      eltx8 rows[8];\n#pragma unroll\n      for (int half = 0; half < 2; ++half) {")
    b = s.index("      float ca[GNB ? 8 : 1], cbf[GNB ? 8 : 1];")
    s = s[:a] + ("      eltx8 rows[8];\n#pragma unroll\n      for (int j = 0; j < 8; ++j)\n#pragma unroll\n"
                 "        for (int k = 0; k < 8; ++k) rows[j][k] = (elt)acc[b][j >> 1][(j & 1) * 8 + k];  // BOUND PROBE: no transposition\n") + s[b:]
    s = s.replace("        } else if constexpr (STATS) {\n          const eltx8 vz = ok ? v : eltx8{};", "        } else if constexpr (STATS && false) {\n          const eltx8 vz = ok ? v : eltx8{};")
elif P == "nobar":
    s = s.replace("            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n            __builtin_amdgcn_s_barrier();", "            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");")
elif P == "zerodma":  # every DMA of the tap loops is issued, but with a resource of 0 bytes: no memory access, zeros land in LDS
    s = s.replace("    rr[2] = real ? rs[2] : 0u;\n    lds_dma(rr, goff[it], (unsigned)kc * 32u, real ?", "    rr[2] = 0u;\n    lds_dma(rr, goff[it], (unsigned)kc * 32u, real ?")
    s = s.replace("    rr[2] = real ? w_bytes : 0u;", "    rr[2] = 0u;")
elif P == "wonly":  # weight DMAs only
    s = s.replace("              if (m == 7 && g == 0 && tap9 < 5) in_dma(", "              if (false) in_dma(").replace(
        "              if (m == 7 && g == 1 && tap9 < 4) in_dma(", "              if (false) in_dma(")
    s = s.replace('if (g == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");', 'if (g == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");')
    s = s.replace('else if (g == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");', 'else if (g == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");')
elif P == "inonly":  # input DMAs only
    s = s.replace("              if (m == 6 && tap9 < 5) w_dma(", "              if (false) w_dma(")
    s = s.replace('if (g == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");', 'if (g == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");')
    s = s.replace('else if (g == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");', 'else if (g == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");')
    s = s.replace('else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");', 'else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
elif P == "nodma":
    s = s.replace("              if (m == 6 && tap9 < 5) w_dma(", "              if (false) w_dma(").replace("              if (m == 7 && g == 0 && tap9 < 5) in_dma(", "              if (false) in_dma(").replace(
        "              if (m == 7 && g == 1 && tap9 < 4) in_dma(", "              if (false) in_dma(")
    s = s.replace('if (g == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");', 'if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
open(W + '/conv2b_mfma.inc', 'w').write(s)
PY
  ( cd $W && /opt/rocm/bin/hipcc $F -c conv_mfma.hip -o conv_mfma.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/api.o $O/head_mfma.o $O/head_mfma_f16.o $O/conv_direct.o \
      conv_mfma.o $O/conv_mfma_f16.o $O/conv_f32_mfma.o $O/conv_x3_mfma.o $O/norm_act.o $O/loss.o $O/head_loss.o $O/predict.o $O/augment.o \
      -o $R/torch-mednet_amd/mednet_hip/libmednet_hip_c2b_$P.so && echo built $P ) &
done
wait
