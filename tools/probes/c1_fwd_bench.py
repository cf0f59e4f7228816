"""First layer (Cin = 1 -> 32 / 64) forward with fused GroupNorm statistics, alone on the chip: HIP events over 20 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import nn as hnn
dev = "cuda:0"
from mednet_hip import _lib as L
for mode, persist in (("bf16", 1), ("bf16", 0), ("fp16", 1), ("fp16x2", 1), ("fp16x2", 0)):
    L.lib().mednet_set_option(b"conv_c1_persist", persist)  # 0: one workgroup per (brick, channel block) item (until round 6)
    for n, cout, shape in ((4, 32, (128, 128, 128)), (2, 64, (160, 160, 96))):
        with mednet_hip.precision(mode):
            conv = hnn.Conv3d(1, cout, 3, bias=False).to(dev)
            x = torch.randn(n, 1, *shape, device=dev)
            for _ in range(3):
                y, p = conv.forward_with_stats(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(20):
                y, p = conv.forward_with_stats(x)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            nbytes = y.numel() * 2 + x.numel() * 4
            print(f"{mode} persist={persist} 1->{cout} @{shape} N={n}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s (Python call included)", flush=True)
