#!/bin/bash
# Round 6: what power and clock does the part hold under each kind of kernel?  rocm-smi sampled every 0.4 s while one kernel loops for
# ~6 s: the bare MFMA probe loops, conv32 (forward + statistics, 128^3 N = 4), the co-resident weight gradient, a GroupNorm apply pass,
# and the whole training step.  Output: gpurun_out/r06_power_probe.log
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
mkdir -p gpurun_out
OUT=gpurun_out/r06_power_probe.log; : > $OUT
sample() {  # $1 = label, runs until the background job $2 ends
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' | sed "s/^/$1 | /" >> $OUT; echo >> $OUT
    sleep 0.4
  done
}
cat > /tmp/loop.py <<'PY'
import os, sys, time, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "torch-mednet_amd")]
import mednet_hip
from mednet_hip import _lib as L, ops
which = sys.argv[1]
dev = "cuda:0"; lib = L.lib(); CL = torch.channels_last_3d
N, s, c = 4, 128, 32
g = torch.Generator(device=dev).manual_seed(1)
x = torch.nn.functional.elu(torch.randn(N, c, s, s, s, device=dev, generator=g)).bfloat16().contiguous(memory_format=CL)
dy = (torch.randn(N, c, s, s, s, device=dev, generator=g) * 1e-3).bfloat16().contiguous(memory_format=CL)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv_weight(w, 3, False)
y = torch.empty_like(x); st = torch.cuda.current_stream().cuda_stream
rows = lib.mednet_conv3d_fused_stats_chunks(N, s, s, s, c, c, 3, 1, 1, 2)
part = torch.empty(N, rows, c, 2, device=dev)
dw = torch.empty(c, c, 3, 3, 3, device=dev)
ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(N, s, s, s, c, c, 3, 0), dtype=torch.uint8, device=dev)
if which == "conv32":
    fn = lambda: lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, c, c, 3, 1, 0, 1, 0, 0, 2, part.data_ptr(), st)
elif which == "wgrad":
    fn = lambda: lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, s, s, s, c, c, 3, 1, 0, 1, 0, 2, 0, ws.data_ptr(), ws.numel(), st)
elif which == "copy":
    fn = lambda: y.copy_(x)
elif which == "step":
    from mednet_hip.train import SegmentationStep
    from mednet_hip.unet.model import ResidualUNet3D
    from mednet_hip.synth import keyed_init_, synthetic_batch
    mednet_hip.set_precision("bf16")
    model = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev)
    stp = SegmentationStep(model, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
    b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=1234).items()}
    fn = lambda: stp(b)
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
print(which, "launches", n, "avg us", (time.time() - t0) / n * 1e6, flush=True)
PY
echo "idle | $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk|fclk' | tr '\n' ' ')" >> $OUT
for k in conv32 wgrad copy step; do
  python /tmp/loop.py $k >> $OUT 2>&1 &
  sample $k $!
done
( for i in 1 2 3 4 5 6; do ./tools/probes/mfma_shape_probe > /dev/null 2>&1; done ) &
sample mfma_probe $!
python - <<'PY'
import re, collections
rows = collections.defaultdict(list)
for l in open("gpurun_out/r06_power_probe.log"):
    if " | " not in l: print(l.strip()); continue
    lab, rest = l.split(" | ", 1)
    p = re.search(r"Power \(W\): ([0-9.]+)", rest); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", rest)
    if p: rows[lab].append((float(p.group(1)), int(c.group(1)) if c else -1))
for lab, v in rows.items():
    v = v[2:] or v
    print(f"{lab:12s} samples {len(v):3d}  power W: mean {sum(a for a, _ in v) / len(v):7.1f} max {max(a for a, _ in v):7.1f}   sclk MHz: mean {sum(b for _, b in v) / len(v):6.0f} min {min(b for _, b in v)}")
PY
