#!/bin/bash
# One GPU-box call: bench line (+ optional rocprof kernel stats of the same command).
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --cpu-steps 0}"
timeout -k 10 900 python bench.py $ARGS > gpurun_out/bench.log 2>&1 || { tail -30 gpurun_out/bench.log; exit 1; }
tail -3 gpurun_out/bench.log
if [ -n "$PROFILE" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -- python3 "$OLDPWD/bench.py" $ARGS --no-roofline > "$OLDPWD/gpurun_out/prof.log" 2>&1
  cd "$OLDPWD"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -30 "$f"
fi
