export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
B=$PWD/torch-mednet_amd/mednet_hip/libmednet_hip_epibound.so
for r in 1 2 3; do
  echo "== product"; python tools/probes/conv32_timing.py 2>&1 | grep "conv32=1 stats"
  echo "== no-epilogue bound"; MEDNET_LIB_PATH=$B python tools/probes/conv32_timing.py 2>&1 | grep "conv32=1 stats"
done
for r in 1 2 3; do
  echo "== product step"; python bench.py --steps 10 --warmup 3 --cpu-steps 0 --fp32-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_ms'])"
  echo "== bound step"; MEDNET_LIB_PATH=$B python bench.py --steps 10 --warmup 3 --cpu-steps 0 --fp32-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_ms'])"
done
