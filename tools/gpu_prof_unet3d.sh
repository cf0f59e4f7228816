#!/bin/bash
# rocprofv3 kernel stats for the UNet3D variant of config 2 (secondary model).
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
export RC_WHICH=${RC_WHICH:-unet3d}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_unet3d
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_unet3d -- python3 $R/tools/run_configs.py > $R/gpurun_out/prof_unet3d.log 2>&1
echo "rc=$?"
tail -2 $R/gpurun_out/prof_unet3d.log
F=$(find $R/gpurun_out/prof_unet3d -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), ("%.3f" % (float(r["TotalDurationNs"]) / 1e6)).rjust(10), "ms", r["Percentage"].rjust(7))
PY
