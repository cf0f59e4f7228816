#!/bin/bash
# Kernel traces of the training step with the side stream off (every kernel alone on the chip): stand-alone durations
# of each kernel of the step, in both storage modes.  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp MEDNET_SIDE_STREAM=0
R=$PWD
for P in bf16 fp32; do
  rm -rf gpurun_out/ss_$P; ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ss_$P -- python3 $R/bench.py --precision $P --steps 4 --warmup 3 \
      --cpu-steps 0 --fp32-steps 0 --no-roofline > $R/gpurun_out/ss_$P.log 2>&1 )
  python3 tools/step_timeline.py $(find gpurun_out/ss_$P -name "*kernel_trace.csv" | head -1) 40 > gpurun_out/ss_${P}_timeline.txt
  tail -30 gpurun_out/ss_${P}_timeline.txt
done
