#!/bin/bash
# round 6, call A: the timed-workload parity tests (live oracle, N = 4) + the bench line with the new fields
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
( rocminfo | grep -m1 -E "gfx9" ; nproc ; free -g | head -2 ) > gpurun_out/box.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_timed_workload.py -m gpu -q -s --tb=short > gpurun_out/r06_timed_workload_parity.log 2>&1
rc=$?
grep -E "timed workload|passed|failed|Error|error" gpurun_out/r06_timed_workload_parity.log | cut -c1-600
echo "pytest rc=$rc"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 > gpurun_out/bench_r06a.log 2>&1 || { tail -30 gpurun_out/bench_r06a.log; exit 1; }
tail -1 gpurun_out/bench_r06a.log > gpurun_out/bench_r06a.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r06a.json"))
print(d["value"], d["ms_per_step"]); print(d["roofline"]); print(d.get("roofline_dgrad")); print(d.get("roofline_wgrad"))
PY
MEDNET_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 --no-roofline > gpurun_out/bench_r06a_dist.log 2>&1 || { tail -30 gpurun_out/bench_r06a_dist.log; exit 1; }
tail -1 gpurun_out/bench_r06a_dist.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d.get('rccl'))"
exit $rc
