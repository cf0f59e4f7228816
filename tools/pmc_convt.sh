#!/bin/bash
# LDS bank conflicts / activity of the ConvTranspose kernels (micro-benchmark)
mkdir -p gpurun_out
R=$PWD
export PYTHONPATH="$R:$R/torch-mednet_amd:$PYTHONPATH"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_ct
KB_WHICH=convt KB_ITERS=2 timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ct -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/pmc_ct.log 2>&1
echo "rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pmc_ct/*/*counter_collection.csv')
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    n = r['Kernel_Name']
    if 'conv_mfma_kernel<2>' in n or 'convt_' in n:
        agg[(n.split('(')[0][-30:], r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(k, len(v), '%.4g' % (sum(v) / len(v)))
PY
