"""Does v_mfma_f32_32x32x16_f16 honour subnormal fp16 INPUTS?  (round 6: the low weight image of the split-weight fp16 mode holds
values around 1e-6, below fp16's smallest normal 6.1e-5.)  A 32 -> 32 convolution with every weight = 3e-6 (fp16 subnormal 0x0032)
on an input of ones: 27 * 32 * 3e-6 = 2.6e-3 per output if the matrix core multiplies subnormals, 0 if it flushes them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import nn as hnn
dev = "cuda:0"
for mode in ("fp16", "bf16"):
    with mednet_hip.precision(mode):
        mednet_hip.set_conv_algo("mfma")
        conv = hnn.Conv3d(32, 32, 3, bias=False).to(dev)
        with torch.no_grad():
            conv.weight.fill_(3e-6)
        x = torch.ones(1, 32, 16, 16, 32, device=dev).to(torch.float16 if mode == 'fp16' else torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
        y = conv(x)
        mednet_hip.set_conv_algo("auto")
    w16 = torch.tensor(3e-6).half() if mode == "fp16" else torch.tensor(3e-6).bfloat16()
    print(mode, "weight as stored:", float(w16), " centre output:", float(y[0, 0, 8, 8, 16]), " expected:", 27 * 32 * float(w16))
