"""Stand-alone timings of the 1x1x1 head's kernels at config-2 / config-4 shapes (N=4, 128^3, 32 channels, 4 or 18 classes):
forward, data gradient (+ GroupNorm-3 sums), weight gradient + bias sums."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import nn as hnn

dev = "cuda:0"
N, S, K = 4, 128, 32
with mednet_hip.precision("bf16"):
    for m in (4, 18):
        head = hnn.Conv3d(K, m, 1, planar_output=True).to(dev)
        x = torch.randn(N, K, S, S, S, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        cot = torch.randn(N, m, S, S, S, device=dev)
        def step():
            x.grad = None
            head.weight.grad = head.bias.grad = None
            y = head(x)
            y.backward(cot)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        print(f"head {K}->{m} @ {S}^3 N={N}: forward + data gradient + weight/bias gradient {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us")
