#!/bin/bash
# Kernel trace of the two-stream training step under a set of option strings: the timeline of the last step shows which kernels
# of the main stream (q=1) and of the weight-gradient stream (q=2) really ran side by side.  Run on the GPU box from the repo root.
# usage: tools/two_stream_trace.sh name1:OPTIONS1 [name2:OPTIONS2 ...]   (OPTIONS as in MEDNET_OPTIONS, may be empty)
set -e
export TMPDIR=/tmp
R=$PWD
for arm in "$@"; do
  name="${arm%%:*}"; opts="${arm#*:}"
  rm -rf gpurun_out/ts_$name
  ( cd /tmp && MEDNET_OPTIONS="$opts" rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ts_$name -- python3 $R/bench.py --steps 4 --warmup 3 \
      --cpu-steps 0 --fp32-steps 0 --no-roofline > $R/gpurun_out/ts_$name.log 2>&1 )
  python3 tools/step_timeline.py $(find gpurun_out/ts_$name -name "*kernel_trace.csv" | head -1) 40 > gpurun_out/ts_${name}_timeline.txt
  rm -rf gpurun_out/ts_$name
  tail -28 gpurun_out/ts_${name}_timeline.txt
done
