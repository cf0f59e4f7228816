#!/bin/bash
# Two ranks of bench.py on ONE GPU over gloo (MEDNET_REHEARSE_ONE_GPU=1): the launcher line the driver uses, both ranks' code
# paths, the bucketed exchange, barriers and the MAX-over-ranks timing -- everything of an N = 2 run except RCCL and the second
# device.  Not a throughput.  Run on the GPU box from the repo root.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH" MEDNET_REHEARSE_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${RANKS:-2} --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus ${RANKS:-2} --steps 6 --warmup 3 --cpu-steps 0 --fp32-steps 0 --no-roofline > gpurun_out/rehearse2.log 2>&1
rc=$?
tail -3 gpurun_out/rehearse2.log | cut -c1-1200
echo "rc=$rc"
exit $rc
