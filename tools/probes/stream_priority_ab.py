"""Does a HIP stream priority change how the weight-gradient stream and the main stream share the chip?  BASELINE config 2's step
(bf16) with (a) both streams at the default priority, (b) the whole step issued on a HIGH-priority stream (the weight gradients stay
on a default-priority side stream), (c) the side stream at the lowest priority the runtime offers.  One process per arm (the side
stream is created once per process); called by tools/gpu_r6m.sh."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip.train import SegmentationStep
from mednet_hip.unet.model import ResidualUNet3D
from mednet_hip.synth import keyed_init_, synthetic_batch

dev = torch.device("cuda:0")
arm = sys.argv[1]
try:
    lo, hi = torch.cuda.Stream.priority_range()
except Exception:
    lo, hi = 0, -1
main = torch.cuda.Stream(device=dev, priority=hi) if arm == "main_high" else torch.cuda.current_stream(dev)
with torch.cuda.stream(main), mednet_hip.precision("bf16"):
    model = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev)
    step = SegmentationStep(model, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
    b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=1234).items()}
    for _ in range(8):
        step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        loss = step(b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
print(f"{arm:10s} priority range (lowest, highest) = ({lo}, {hi})  side priority {os.environ.get('MEDNET_SIDE_PRIORITY', '0')}: {ms:.3f} ms per step, loss {float(loss):.5f}", flush=True)
