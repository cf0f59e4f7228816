export TMPDIR=/tmp MEDNET_SIDE_STREAM=0 RC_WHICH=unet3d RC_PREC=bf16
R=$PWD
rm -rf gpurun_out/ss_u3d; ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ss_u3d -- python3 $R/tools/run_configs.py > $R/gpurun_out/ss_u3d.log 2>&1 )
python3 tools/step_timeline.py $(find gpurun_out/ss_u3d -name "*kernel_trace.csv" | head -1) 100 > gpurun_out/ss_u3d_timeline.txt
tail -32 gpurun_out/ss_u3d_timeline.txt
