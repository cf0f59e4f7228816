#!/bin/bash
# round 6, call D: conv32 with row-pair fragments: the conv tests, its stand-alone timing, the step
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -k "conv32 or two_block or conv3d_mfma or narrow or conv" > gpurun_out/r06_conv32_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r06_conv32_tests.log | cut -c1-300
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/probes/conv32_timing.py 2>&1 | tee gpurun_out/r06_conv32_timing.log
python bench.py --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 > gpurun_out/bench_r06d.log 2>&1 || { tail -20 gpurun_out/bench_r06d.log; exit 1; }
tail -1 gpurun_out/bench_r06d.log > gpurun_out/bench_r06d.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r06d.json"))
print(d["value"], d["ms_per_step"]); print({k: d["roofline"][k] for k in ("achieved", "frac", "avg_ms", "family")}); print(d.get("roofline_dgrad", {}).get("variants"))
PY
