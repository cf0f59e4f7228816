#!/bin/bash
# Register / scratch / LDS usage of every kernel as hipcc reports it (-Rpass-analysis=kernel-resource-usage), one line per
# kernel -> profiles/${ROUND}_kernel_resource_usage.txt.  Runs on the build container (cross-compiles, no GPU).
ROUND=${ROUND:-r04}
cd "$(dirname "$0")/../torch-mednet_amd/csrc" || exit 1
OUT=../../profiles/${ROUND}_kernel_resource_usage.txt
echo "# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, $(git rev-parse --short HEAD 2>/dev/null)" > $OUT
for f in conv_mfma.hip conv_x3_mfma.hip conv_f32_mfma.hip conv_direct.hip norm_act.hip loss.hip head_loss.hip predict.hip augment.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/ru.o 2>&1 |
  python3 -c "
import re,sys,subprocess
txt=sys.stdin.read().split('Function Name: ')[1:]
for t in txt:
    name=t.split(' ')[0]
    g=lambda k: (re.search(k+r'[^:\n]*: (\d+)',t) or [0,'?'])[1]
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    dem=re.sub(r'\(.*','',dem)[:70]
    print(f'$f  {dem:70s} VGPRs={g(\"VGPRs\")} AGPRs={g(\"AGPRs\")} SGPRs={g(\"SGPRs\")} ScratchSize={g(\"ScratchSize\")} VGPRSpill={g(\"VGPRs Spill\")} Occupancy={g(\"Occupancy\")} LDS={g(\"LDS Size\")}')
" >> $OUT
done
grep -c . $OUT; awk '{for(i=1;i<=NF;i++) if ($i ~ /^ScratchSize=/ && $i != "ScratchSize=0") print}' $OUT
