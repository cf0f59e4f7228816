#!/bin/bash
# HBM traffic of the fp32-mode step's kernels: separate FETCH_SIZE / WRITE_SIZE passes of `bench.py --precision fp32` (program
# directly after `--`), summarised per launch slot by tools/pmc_fp32_summarize.py.  Run on the GPU box from the repo root.
set -o pipefail
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
R=$PWD
cd /tmp && export TMPDIR=/tmp
PARGS="--precision fp32 --steps 1 --warmup 1 --cpu-steps 0 --fp32-steps 0 --no-roofline"
rm -rf $R/gpurun_out/pmc32_fetch $R/gpurun_out/pmc32_write
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc32_fetch -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc32_fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc32_write -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc32_write.log 2>&1
echo "write rc=$?"
cd $R && python3 tools/pmc_fp32_summarize.py
