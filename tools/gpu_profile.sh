#!/bin/bash
# One GPU-box call: full bench line (with cpu_baseline), rocprofv3 kernel stats, and the two PMC passes (FETCH_SIZE,
# WRITE_SIZE) for the dominant kernel's HBM traffic.  Outputs under gpurun_out/ (copy what is judged into profiles/).
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
R=$PWD
timeout -k 10 600 python bench.py ${BENCH_FULL_ARGS:-} > gpurun_out/bench_full.log 2>&1 || { tail -30 gpurun_out/bench_full.log; exit 1; }
tail -2 gpurun_out/bench_full.log
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --cpu-steps 0 --no-roofline"
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats.log 2>&1
echo "stats rc=$?"
PARGS="--steps 1 --warmup 1 --cpu-steps 0 --no-roofline"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc_write.log 2>&1
echo "pmc write rc=$?"
cd $R
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" | head -20
du -sh gpurun_out
