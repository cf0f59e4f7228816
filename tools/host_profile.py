"""Where does the host spend its ~15 ms per step?  cProfile over 5 eager steps of the bench workload."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip.synth import keyed_init_, synthetic_batch
from mednet_hip.train import SegmentationStep
from mednet_hip.unet.model import ResidualUNet3D

dev = torch.device("cuda", 0)
mednet_hip.set_precision("bf16")
net = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev)
step = SegmentationStep(net, loss_weight=[0.05, 1, 1, 1.0], graph=False)
b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=1).items()}
for _ in range(3):
    step(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
