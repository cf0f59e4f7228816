"""First-layer kernel timed INSIDE the training step (HIP events around its launch, ops.PROFILE) vs stand-alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch, mednet_hip
from mednet_hip import ops
from mednet_hip.train import SegmentationStep
from mednet_hip.unet.model import ResidualUNet3D
from mednet_hip.synth import keyed_init_, synthetic_batch
dev = torch.device("cuda", 0)
mednet_hip.set_precision("fp32")
model = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev)
step = SegmentationStep(model, loss_weight=[0.05, 1, 1, 1.0], lr=1e-3)
b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=1234).items()}
for _ in range(3):
    step(b)
torch.cuda.synchronize()
ops.PROFILE.update(enabled=True, events=[], match=lambda k, ci, co, d, h, w: ci == 1)
for _ in range(4):
    step(b)
torch.cuda.synchronize()
ops.PROFILE["enabled"] = False
print("in-step c1 launches (us):", [round(e0.elapsed_time(e1) * 1e3, 1) for e0, e1, _ in ops.PROFILE["events"]])
# the same forward alone, same tensors
x = b["data"].float()
with torch.no_grad():
    enc = model.encoders[0].basic_module
    conv = enc.conv1.conv
    for _ in range(3):
        y, p = conv.forward_with_stats(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        y, p = conv.forward_with_stats(x)
    e1.record()
    torch.cuda.synchronize()
print("alone, through the module: %.1f us" % (e0.elapsed_time(e1) * 100))
print("x: ", x.shape, x.stride(), x.dtype, " weight", conv.weight.shape)
