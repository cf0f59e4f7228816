#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the conv micro-benchmark launches, persistent vs one-workgroup-per-item
mkdir -p gpurun_out
R=$PWD
export PYTHONPATH="$R:$R/torch-mednet_amd:$PYTHONPATH"
cd /tmp && export TMPDIR=/tmp
for P in 0 1; do
  rm -rf $R/gpurun_out/pmc_fc$P
  MEDNET_OPTIONS=conv_persist=$P KB_WHICH=conv KB_ITERS=2 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fc$P -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/pmc_fc$P.log 2>&1
  echo "persist=$P rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for P in (0, 1):
    fs = glob.glob(f'gpurun_out/pmc_fc{P}/*/*counter_collection.csv')
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'conv_mfma_kernel<1>' in r['Kernel_Name']:
            agg[(r['Grid_Size'], round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e5))].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print('persist', P, 'grid', k[0], 'dur~%d00us' % k[1], len(v), 'FETCH_SIZE KB avg %.0f' % (sum(v) / len(v)))
PY
