# Scan of wgrad_mfma4_kernel's ring depth (compile-time: hipcc -DMEDNET_WG4_DEPTH=2|4 -c conv_mfma.hip, linked as libmednet_hip_wg4d2.so / _wg4d4.so
# next to the product library) and z-slab depth (option wgrad4_zs): profiles/r05_wgrad4_depth_slab_scan.log -- all within 1.5 %.
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
M=$PWD/torch-mednet_amd/mednet_hip
for r in 1 2; do
for v in "d3zs32:" "d3zs64:wgrad4_zs=64" "d3zs16:wgrad4_zs=16" "d2zs32:LIB=$M/libmednet_hip_wg4d2.so" "d4zs32:LIB=$M/libmednet_hip_wg4d4.so" "d2zs64:LIB=$M/libmednet_hip_wg4d2.so,wgrad4_zs=64"; do
  name=${v%%:*}; rest=${v#*:}; lib=""; opts=""
  IFS=',' read -ra KV <<< "$rest"
  for kv in "${KV[@]}"; do case $kv in LIB=*) lib=${kv#LIB=};; "") ;; *) opts="$opts,$kv";; esac; done
  echo "== $name"
  MEDNET_LIB_PATH=$lib MEDNET_OPTIONS=${opts#,} WG_SHAPES=32x32x128,64x64x64 python tools/probes/wgrad4_bench.py 2>&1 | grep -v "amdgpu.ids" | grep "^wgrad\|v4 wgs=256"
done
done
