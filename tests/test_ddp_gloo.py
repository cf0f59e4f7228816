"""The N>1 path on CPU: world_size-2 gloo.  Checks the trainer's flat-buffer exchange (train.FlatParams +
train.allreduce_gradients) against the definition in SURVEY 8(e): every rank ends with the MEAN of the per-rank-batch
gradients (Dice is reduced over the local batch first), and identical parameters after the update."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "torch-mednet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ref_cpu as O
    from mednet_hip.train import FlatParams, allreduce_gradients
    model = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8]))
    flat = FlatParams(model)
    # parameters now alias the flat buffer
    assert all(p.data.data_ptr() >= flat.flat.data_ptr() for p in flat.params)
    batch = O.synthetic_batch(2, 1, (16, 16, 16), 2, 0, seed=1234 + rank)  # per-rank patches (seed + rank)
    loss = O.seg_training_step(model, O.DiceLoss(weight=torch.tensor([0.05, 1.0])), batch)
    loss.backward()
    for p in flat.params:  # the CPU oracle has no direct-gradient kernels: stage its grads into the flat buffer
        p._mednet_grad.copy_(p.grad)
    local = flat.grad.clone()
    scale = allreduce_gradients(flat.grad, world)
    torch.save({"local": local, "reduced": flat.grad * scale, "scale": scale}, os.path.join(out, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["scale"] == 0.5
    assert not torch.equal(r0["local"], r1["local"])  # different patches per rank
    mean = 0.5 * (r0["local"] + r1["local"])
    assert torch.allclose(r0["reduced"], mean, rtol=1e-6, atol=1e-9)
    assert torch.equal(r0["reduced"], r1["reduced"])  # every rank applies the same update


def _cpu_hybrid_of_product_model(O):
    """mednet_hip's ResidualUNet3D skeleton with the oracle's CPU leaf modules plugged in (same parameter order), so the
    product's own forward control flow runs without a GPU."""
    import torch.nn as nn
    from mednet_hip.unet import model as HM

    class UpWithSkip(nn.Module):  # the product's Decoder calls upsample(x, skip=...) (ConvTranspose3d + skip in one kernel)
        def __init__(self, up):
            super().__init__()
            self.up = up

        def forward(self, x, skip=None):
            return self.up(x) + skip

    ora = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8, 16, 32]))
    net = HM.ResidualUNet3D(1, 2, False, f_maps=[8, 16, 32])
    for he, oe in zip(net.encoders, ora.encoders):
        he.basic_module, he.pooling = oe.basic_module, oe.pooling
    for hd, od in zip(net.decoders, ora.decoders):
        hd.basic_module, hd.upsample = od.basic_module, UpWithSkip(od.upsample)
    net.final_conv = ora.final_conv
    return net


def _bucket_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "torch-mednet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MEDNET_BUCKETS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ref_cpu as O
    from mednet_hip.train import FlatParams, BucketedExchange, late_bucket_split
    batch = O.synthetic_batch(1, 1, (16, 16, 16), 2, 0, seed=77 + rank)
    crit = O.DiceLoss(weight=torch.tensor([0.05, 1.0]))
    # plain local gradients
    ref = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8, 16, 32]))
    O.seg_training_step(ref, crit, batch).backward()
    local = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    # bucketed exchange: gradients accumulate straight into the flat buffer (p.grad aliases its slice)
    model = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8, 16, 32]))
    flat = FlatParams(model)
    ex = BucketedExchange(model, flat, world)
    split = late_bucket_split(model, flat)
    assert ex.enabled and 0 < split < flat.total
    n_late = sum(p.numel() for enc in list(model.encoders)[:2] for p in enc.parameters())
    assert split >= n_late and split - n_late < 64 * len(flat.params)  # (slices are padded to 64 elements)
    flat.grads_as_attr()
    fired = []
    ex_on_grad = ex._on_grad
    ex._on_grad = lambda g: (fired.append(1), ex_on_grad(g))[1]
    O.seg_training_step(model, crit, batch).backward()
    assert fired and ex.work is not None  # the early bucket was launched from inside backward
    scale = ex.finish()
    got = torch.cat([flat.grad[o:o + p.numel()] for p, o in zip(flat.params, flat.offsets)]) * scale
    # The product model drives its encoders through Encoder.forward(x, with_skip=True) (-> a tuple), not enc(x): the same
    # exchange on the product's module skeleton (_UNetCore.forward, Encoder, Decoder) over CPU leaf blocks must fire too.
    hyb = _cpu_hybrid_of_product_model(O)
    flat2 = FlatParams(hyb)
    ex2 = BucketedExchange(hyb, flat2, world)
    assert ex2.enabled
    flat2.grads_as_attr()
    fired2 = []
    ex2_on_grad = ex2._on_grad
    ex2._on_grad = lambda g: (fired2.append(1), ex2_on_grad(g))[1]
    crit(hyb(batch["data"].float()), batch["label"][:, -1].long()).backward()
    assert fired2 and ex2.work is not None, "BucketedExchange did not fire on the product model's forward path"
    scale2 = ex2.finish()
    got2 = torch.cat([flat2.grad[o:o + p.numel()] for p, o in zip(flat2.params, flat2.offsets)]) * scale2
    torch.save({"local": local, "reduced": got, "reduced_product_skeleton": got2}, os.path.join(out, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_exchange_overlapped_with_backward_world2(tmp_path):
    """train.BucketedExchange: the early bucket (everything but the first two encoders) is all-reduced from a gradient
    hook INSIDE backward, the rest afterwards; the result is the mean of the per-rank gradients on every rank."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "b0.pt")
    r1 = torch.load(tmp_path / "b1.pt")
    mean = 0.5 * (r0["local"] + r1["local"])
    assert torch.allclose(r0["reduced"], mean, rtol=1e-6, atol=1e-9)
    assert torch.equal(r0["reduced"], r1["reduced"])
    assert torch.allclose(r0["reduced_product_skeleton"], mean, rtol=1e-5, atol=1e-8)
    assert torch.equal(r0["reduced_product_skeleton"], r1["reduced_product_skeleton"])


def test_synth_generators_match_oracle():
    from oracle import ref_cpu as O
    from mednet_hip import synth
    a = O.synthetic_batch(2, 1, (8, 8, 8), 4, 3, seed=99)
    b = synth.synthetic_batch(2, 1, (8, 8, 8), 4, 3, seed=99)
    assert torch.equal(a["data"], b["data"]) and torch.equal(a["label"], b["label"])
    m1 = O.keyed_init_(O.ResidualUNet3D(1, 4, False, f_maps=[8, 16]))
    m2 = synth.keyed_init_(O.ResidualUNet3D(1, 4, False, f_maps=[8, 16]))
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), k
