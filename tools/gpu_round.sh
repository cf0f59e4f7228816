set -o pipefail
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
bash tools/gpu_check.sh || exit 1
for o in "conv_narrow=1" "conv_narrow=0" "conv_narrow=1"; do
  echo "== $o"; MEDNET_OPTIONS=$o RC_WHICH=cfg5only RC_PREC=bf16 python tools/run_configs.py 2>&1 | tail -1 | cut -c1-200
done
python bench.py --steps 20 --warmup 5 --cpu-steps 0 --fp32-steps 0 2>&1 | tail -1 > gpurun_out/bench_r05_dev.json; cut -c1-400 gpurun_out/bench_r05_dev.json; python -c "
import json; d=json.load(open('gpurun_out/bench_r05_dev.json')); print(d['roofline']); print(d.get('roofline_wgrad'))"
