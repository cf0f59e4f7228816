#!/bin/bash
# Second GPU-box call of the profile set (after tools/gpu_profile.sh + tools/pmc_summarize.py have refreshed the PMC traffic
# file): the bench line with `traffic` filled in, the other configurations, single-stream and two-stream timelines of both modes.
set -o pipefail
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
R=$PWD
python bench.py > gpurun_out/bench_line_final.log 2>&1 && tail -1 gpurun_out/bench_line_final.log | cut -c1-200
python tools/run_configs.py > gpurun_out/configs_final.log 2>&1; grep -c . gpurun_out/configs_final.log
bash tools/single_stream_trace.sh > /dev/null 2>&1
export TMPDIR=/tmp
for P in bf16 fp32; do
  rm -rf gpurun_out/ts_$P
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ts_$P -- python3 $R/bench.py --precision $P --steps 5 --warmup 3 --cpu-steps 0 --fp32-steps 0 --no-roofline > $R/gpurun_out/ts_$P.log 2>&1 )
  python3 tools/step_timeline.py $(find gpurun_out/ts_$P -name "*kernel_trace.csv" | head -1) 150 > gpurun_out/ts_${P}_timeline.txt
  tail -1 gpurun_out/ts_$P.log | cut -c1-160
done
