#!/bin/bash
# round 6: duration of the ConvTranspose data-gradient launch INSIDE the two-stream step, ticket walk vs static brick lists
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for arm in dyn static; do
  rm -rf $R/gpurun_out/ct32_$arm
  if [ $arm = static ]; then export MEDNET_OPTIONS=convt_dgrad32_dynamic=0; else unset MEDNET_OPTIONS; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ct32_$arm -- python3 $R/bench.py --steps 5 --warmup 3 --cpu-steps 0 --fp32-steps 0 --no-roofline > $R/gpurun_out/ct32_$arm.log 2>&1
  python3 $R/tools/step_timeline.py $(find $R/gpurun_out/ct32_$arm -name "*kernel_trace.csv" | head -1) 150 > $R/gpurun_out/ct32_${arm}_timeline.txt
  echo "== $arm"; grep -n "convt_dgrad32" -B4 -A3 $R/gpurun_out/ct32_${arm}_timeline.txt | head -24
  grep "convt_dgrad32" $(find $R/gpurun_out/ct32_$arm -name "*kernel_stats.csv" | head -1)
done
