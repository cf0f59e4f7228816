#!/bin/bash
# round 6: stream priorities of the main and the weight-gradient stream, three interleaved rounds
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
for i in 1 2 3; do
  python tools/probes/stream_priority_ab.py default
  python tools/probes/stream_priority_ab.py main_high
  MEDNET_SIDE_PRIORITY=1 python tools/probes/stream_priority_ab.py side_low
  MEDNET_SIDE_PRIORITY=-1 python tools/probes/stream_priority_ab.py side_high
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_stream_priority_ab.log
