#!/bin/bash
# round 6: the whole -m gpu suite, then the bench line (no CPU / fp32 sub-records)
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --tb=short > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 > gpurun_out/bench_r06e.log 2>&1 || { tail -20 gpurun_out/bench_r06e.log; exit 1; }
tail -1 gpurun_out/bench_r06e.log > gpurun_out/bench_r06e.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r06e.json"))
print(d["value"], d["ms_per_step"]); print({k: d["roofline"][k] for k in ("achieved", "frac", "avg_ms", "family")})
print({k: (v["avg_ms"], v["frac"]) for k, v in d.get("roofline_dgrad", {}).get("variants", {}).items()}); print(d.get("roofline_wgrad"))
PY
${EXTRA:-true}
