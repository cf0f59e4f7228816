#!/bin/bash
# After tools/gpu_profile.sh + tools/pmc_summarize.py + tools/round_artifacts.sh have run on the GPU box and their outputs have been
# merged into gpurun_out/: copy what is judged into profiles/ under the round's prefix, then write BENCH.md.
# usage: ROUND=r04 tools/collect_profiles.sh
set -e
R=${ROUND:?set ROUND=rNN}
tail -1 gpurun_out/bench_line_final.log > profiles/${R}_bench_line.json
python3 -c "import json,sys; json.load(open('profiles/${R}_bench_line.json'))"
cp gpurun_out/bench_full.log profiles/${R}_bench_full.log
python3 -c "
import json
rows = [json.loads(l) for l in open('gpurun_out/configs_final.log') if l.startswith('{')]
json.dump(rows, open('profiles/${R}_configs.json', 'w'), indent=1)
print(len(rows), 'config lines')"
cp gpurun_out/ss_bf16_timeline.txt profiles/${R}_bf16_single_stream_timeline.txt
cp gpurun_out/ss_fp32_timeline.txt profiles/${R}_fp32_mode_single_stream_timeline.txt
cp gpurun_out/ts_bf16_timeline.txt profiles/${R}_bf16_step_timeline.txt
cp gpurun_out/ts_fp32_timeline.txt profiles/${R}_fp32_mode_step_timeline.txt
ROUND=$R python3 tools/make_bench_md.py
