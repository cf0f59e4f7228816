#!/bin/bash
# round 6: config 5 in fp16 storage + loss scaling against bf16, single-stream kernel traces side by side
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH" TMPDIR=/tmp MEDNET_SIDE_STREAM=0 RC_WHICH=cfg5only
R=$PWD
mkdir -p gpurun_out
for P in fp16 bf16; do
  rm -rf gpurun_out/c5$P
  ( cd /tmp && RC_PREC=$P rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5$P -- python3 $R/tools/run_configs.py > $R/gpurun_out/c5$P.log 2>&1 )
  python3 - "$P" <<'PY'
import csv, glob, sys
p = sys.argv[1]
f = glob.glob(f"gpurun_out/c5{p}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 7
print(f"== {p}: kernel time per step (7 steps traced), top 40")
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"]) / steps / 1e6
for r in rows[:40]:
    print(f"  {r['Name'][:70]:70s} {int(r['Calls']) / steps:6.1f} calls  {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms")
print(f"  sum {tot:.2f} ms")
PY
  tail -1 gpurun_out/c5$P.log | cut -c1-200
  rm -rf gpurun_out/c5$P
done 2>&1 | tee gpurun_out/r06_cfg5_fp16_vs_bf16_kernels.txt
