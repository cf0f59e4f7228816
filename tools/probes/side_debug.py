import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import ops
from mednet_hip.synth import keyed_init_, synthetic_batch
from mednet_hip.train import FlatParams, finish_backward
from mednet_hip.unet.model import ResidualUNet3D
from mednet_hip.unet import loss as HL
dev = "cuda:0"
ctor = dict(in_channels=1, out_channels=5, final_sigmoid=False, f_maps=[8, 16])
batch = {k: v.to(dev) for k, v in synthetic_batch(2, 1, (16, 16, 16), 2, 3, seed=4321).items()}
res = {}
for side in (False, True, False, True):
    ops.SIDE["enabled"] = side
    with mednet_hip.precision("fp32"):
        net = keyed_init_(ResidualUNet3D(**ctor)).to(dev)
        flat = FlatParams(net)
        for rep in range(2):
            flat.grad.fill_(float("nan"))
            out = net(batch["data"].float())
            nh = 3
            loss = HL.DiceLoss(weight=torch.tensor([0.05, 1.0], device=dev)).to(dev)(out[:, nh:], batch["label"][:, -1].long()) + \
                HL.HeatmapRegressionLoss([0.015, 0.02, 0.001], "L2").to(dev)(out[:, :nh], batch["label"][:, :-1])
            loss.backward()
            finish_backward()
            torch.cuda.synchronize()
            g = {k: p._mednet_grad.clone() for k, p in net.named_parameters()}
            res.setdefault(side, []).append(g)
ref = res[False][0]
for side in (False, True):
    for rep, g in enumerate(res[side]):
        bad = [(k, float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30))) for k in ref if not torch.equal(g[k], ref[k])]
        nan = [k for k in g if torch.isnan(g[k]).any()]
        print("side", side, "rep", rep, "differs:", bad[:6], "nan:", nan[:6])
