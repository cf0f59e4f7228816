#!/bin/bash
# PMC pass over the conv micro-benchmark: where do the waves of conv_mfma_kernel spend their cycles?
set -o pipefail
mkdir -p gpurun_out
R=$PWD
export PYTHONPATH="$R:$R/torch-mednet_amd:$PYTHONPATH"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_sq2
KB_WHICH=conv KB_ITERS=2 timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/pmc_sq.log 2>&1
echo "rc=$?"
KB_WHICH=conv KB_ITERS=2 timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/pmc_sq2.log 2>&1
echo "rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc_sq', 'pmc_sq2'):
    fs = glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')
    if not fs: print(d, 'no csv'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        n = r['Kernel_Name']
        if 'conv_mfma_kernel' in n or 'wgrad_mfma_kernel' in n:
            agg[(n.split('(')[0][-34:], r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(d, k, len(v), '%.4g' % (sum(v) / len(v)))
PY
