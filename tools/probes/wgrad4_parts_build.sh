#!/bin/bash
# Round 6: where does wgrad_mfma4_kernel's time go?  Throwaway libraries from patched COPIES of the sources (outputs WRONG):
#   libmednet_hip_wg4_noread.so -- no transposing operand reads in the k-step loop (stale fragments)
#   libmednet_hip_wg4_nodma.so  -- no LDS-DMA of new planes (stale planes)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/torch-mednet_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -mllvm -pragma-unroll-threshold=262144"
for P in ${PARTS:-noread nodma}; do
  W=/tmp/wg4_$P; rm -rf $W; mkdir -p $W; cp $O/*.hip $O/*.h $O/*.inc $W/
  python3 - "$R" "$W" "$P" <<'PY'
import sys
R, W, P = sys.argv[1:4]
c = open(W + '/common.h').read().replace('#include "../../include/mednet_hip.h"', f'#include "{R}/include/mednet_hip.h"')
open(W + '/common.h', 'w').write(c)
s = open(W + '/conv_mfma.hip').read()
if P == "noread":
    old = "        for (int idx = first_rd[i]; idx < first_rd[i + 1]; ++idx) rd(nxt, idx, nks);"
    assert old in s
    s = s.replace(old, "        if (T < 4) for (int idx = first_rd[i]; idx < first_rd[i + 1]; ++idx) rd(nxt, idx, nks);")
elif P == "nodma":
    old = "        if (i == 6 && ks < 5) dma(ks, live);"
    assert old in s
    s = s.replace(old, "        if (i == 6 && ks < 5) dma(ks, live && T < 6);")
open(W + '/conv_mfma.hip', 'w').write(s)
PY
  ( cd $W && /opt/rocm/bin/hipcc $F -c conv_mfma.hip -o conv_mfma.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/api.o $O/head_mfma.o $O/head_mfma_f16.o $O/conv_direct.o \
      conv_mfma.o $O/conv_mfma_f16.o $O/conv_f32_mfma.o $O/conv_x3_mfma.o $O/norm_act.o $O/loss.o $O/head_loss.o $O/predict.o $O/augment.o \
      -o $R/torch-mednet_amd/mednet_hip/libmednet_hip_wg4_$P.so && echo built $P ) &
done
wait
