"""PCIe-inclusive note for DESIGN.md: time to hand one config-2 batch (data fp32 4x1x128^3 + labels int64 4x128^3) from
pinned host memory to HBM, to be set beside ms_per_step (the bench's timed region starts with inputs resident)."""
import time, torch
dev = torch.device("cuda", 0)
x = torch.randn(4, 1, 128, 128, 128).pin_memory()
y = torch.randint(0, 4, (4, 128, 128, 128)).pin_memory()
for _ in range(3):
    x.to(dev, non_blocking=True); y.to(dev, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    a = x.to(dev, non_blocking=True); b = y.to(dev, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
mb = (x.numel() * 4 + y.numel() * 8) / 1e6
print(f"H2D config-2 batch: {mb:.1f} MB in {dt * 1e3:.3f} ms = {mb / dt / 1e3:.1f} GB/s")
