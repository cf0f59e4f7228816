#!/bin/bash
# Kernel trace of config 4 (landmark path: 16 heat maps + 2 classes, 128^3, N = 4) in bf16 storage, two streams as benchmarked:
# where its step differs from config 2's (profiles/rNN_cfg4_bf16_step_timeline.txt).  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp RC_WHICH=cfg4 RC_PREC=bf16
R=$PWD
rm -rf gpurun_out/c4
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c4 -- python3 $R/tools/run_configs.py > $R/gpurun_out/c4.log 2>&1 )
python3 tools/step_timeline.py $(find gpurun_out/c4 -name "*kernel_trace.csv" | head -1) ${MIN_US:-15} > gpurun_out/cfg4_bf16_timeline.txt
rm -rf gpurun_out/c4
tail -32 gpurun_out/cfg4_bf16_timeline.txt
tail -2 gpurun_out/c4.log
