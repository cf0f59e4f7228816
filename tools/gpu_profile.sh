#!/bin/bash
# One GPU-box call: full bench line (with fp32_parity_mode and cpu_baseline), rocprofv3 kernel stats, the two HBM PMC passes
# (FETCH_SIZE, WRITE_SIZE) and one SQ pass (MFMA busy, LDS conflicts / waits) of the SAME command, the program directly
# after `--` (no env/bash hop).  Outputs under gpurun_out/; tools/pmc_summarize.py + tools/make_bench_md.py copy what is
# judged into profiles/ (ROUND=r02 ...) and write BENCH.md.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
R=$PWD
timeout -k 10 900 python bench.py ${BENCH_FULL_ARGS:-} > gpurun_out/bench_full.log 2>&1 || { tail -30 gpurun_out/bench_full.log; exit 1; }
tail -2 gpurun_out/bench_full.log
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --cpu-steps 0 --fp32-steps 0 --no-roofline"
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats.log 2>&1
echo "stats rc=$?"
PARGS="--steps 1 --warmup 1 --cpu-steps 0 --fp32-steps 0 --no-roofline"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc_write.log 2>&1
echo "pmc write rc=$?"
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc_sq.log 2>&1
echo "pmc sq rc=$?"
cd $R
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq -name "*.csv" | head -20
du -sh gpurun_out
