#!/bin/bash
# Kernel trace of config 2's step in fp16 storage (loss scaling on the device): what it adds to the bf16 step.
set -e
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/ts_fp16
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ts_fp16 -- python3 $R/bench.py --precision fp16 --steps 4 --warmup 3 \
    --cpu-steps 0 --fp32-steps 0 --no-roofline > $R/gpurun_out/ts_fp16.log 2>&1 )
python3 tools/step_timeline.py $(find gpurun_out/ts_fp16 -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/ts_fp16_timeline.txt
rm -rf gpurun_out/ts_fp16
tail -45 gpurun_out/ts_fp16_timeline.txt
tail -1 gpurun_out/ts_fp16.log | cut -c1-200
