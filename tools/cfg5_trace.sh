#!/bin/bash
# Kernel trace of config 5 (5 levels, 64 base channels, 160x160x96, N = 2) in bf16 storage with the side stream off: the stand-alone
# duration of every kernel of the step (profiles/rNN_cfg5_bf16_single_stream_timeline.txt).  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp MEDNET_SIDE_STREAM=${MEDNET_SIDE_STREAM:-0} RC_WHICH=cfg5only RC_PREC=bf16
R=$PWD
rm -rf gpurun_out/c5
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c5 -- python3 $R/tools/run_configs.py > $R/gpurun_out/c5.log 2>&1 )
python3 tools/step_timeline.py $(find gpurun_out/c5 -name "*kernel_trace.csv" | head -1) ${MIN_US:-20} > gpurun_out/cfg5_bf16_timeline.txt
rm -rf gpurun_out/c5
tail -32 gpurun_out/cfg5_bf16_timeline.txt
cat gpurun_out/c5.log | tail -3
