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


def _free_port():
    """A port nobody listens on right now (fixed ports collided with other runs on the same host)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


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
    port = _free_port()
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
    port = _free_port()
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


def _torch_adam_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """CPU stand-in for the fused Adam launch (ops.adam_step_ -> mednet_adam_step): torch.optim.Adam's update on the flat
    buffers, with the gradient scale (1/world) the kernel folds in."""
    g = g * grad_scale
    if weight_decay:
        g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (v / (1 - beta2 ** step)).sqrt_().add_(eps)
    p.addcdiv_(m / (1 - beta1 ** step), denom, value=-lr)


def _step_worker(rank, world, port, out, buckets):
    for p in (ROOT, os.path.join(ROOT, "torch-mednet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MEDNET_BUCKETS=buckets)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ref_cpu as O
    from mednet_hip import ops
    from mednet_hip.train import SegmentationStep
    ops.adam_step_ = _torch_adam_step_  # the one device launch of __call__ that has no CPU form
    w = [0.05, 1.0]
    model = _cpu_hybrid_of_product_model(O)
    step = SegmentationStep(model, loss_weight=w, lr=1e-3, world_size=world)
    step.loss = O.DiceLoss(weight=torch.tensor(w))  # (the product's DiceLoss is a HIP kernel)
    step.flat.grads_as_attr()  # CPU leaves: autograd accumulates straight into the flat gradient slices
    losses = []
    for it in range(3):
        step.flat.grad.zero_()
        batch = O.synthetic_batch(1, 1, (16, 16, 16), 2, 0, seed=500 + 10 * it + rank)
        losses.append(float(step(batch)))
    assert step._exchange.enabled == (buckets == "1")
    torch.save({"params": step.flat.flat.clone(), "losses": losses, "t": step.opt.t,
                "exchange": step._exchange.describe()}, os.path.join(out, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("buckets", ["0", "1"])
def test_segmentation_step_call_world2(tmp_path, buckets):
    """SegmentationStep.__call__ end to end on two gloo ranks: forward/backward of the product's module skeleton, the
    gradient exchange (single all-reduce, or the two-bucket overlapped form), 1/world folded into the Adam update, three
    steps.  Both ranks must hold identical parameters, equal to a one-process run that averages the two ranks' gradients
    (Dice over the LOCAL batch, then the mean of the gradients) and steps torch.optim.Adam."""
    port = _free_port()
    mp.spawn(_step_worker, args=(2, port, str(tmp_path), buckets), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "s0.pt")
    r1 = torch.load(tmp_path / "s1.pt")
    assert r0["t"] == r1["t"] == 3
    assert torch.equal(r0["params"], r1["params"])
    assert ("2 buckets" in r0["exchange"]) == (buckets == "1")
    from oracle import ref_cpu as O
    from mednet_hip.train import FlatParams
    ref = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8, 16, 32]))
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    crit = O.DiceLoss(weight=torch.tensor([0.05, 1.0]))
    for it in range(3):
        grads = []
        for rank in range(2):
            opt.zero_grad()
            O.seg_training_step(ref, crit, O.synthetic_batch(1, 1, (16, 16, 16), 2, 0, seed=500 + 10 * it + rank)).backward()
            grads.append([p.grad.clone() for p in ref.parameters()])
        for p, g0, g1 in zip(ref.parameters(), *grads):
            p.grad = 0.5 * (g0 + g1)
        opt.step()
    flat = FlatParams(ref)  # same flat order as the ranks'
    assert torch.allclose(r0["params"], flat.flat, rtol=2e-5, atol=2e-7)


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it must BE the launcher: two child ranks, each with its own
    RANK/LOCAL_RANK and WORLD_SIZE=2 (examples/train_seg.py:126 Trainer(gpus=N)).  --dry-launch stops before any GPU call."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(x["rank"] for x in recs) == [0, 1]
    assert all(x["world_size"] == 2 and x["local_rank"] == x["rank"] and x["master"].startswith("127.0.0.1:") for x in recs)
    assert len({x["pid"] for x in recs}) == 2


def test_bench_world_size_mismatch_is_an_error():
    r = _run_bench(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    r = _run_bench(["--gpus", "1", "--dry-launch"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the loud failure on a box WITHOUT GPUs")
def test_bench_gpus2_without_gpus_fails_inside_the_ranks():
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    # (the launcher ends the other rank as soon as one has failed: usually both messages are there, at least one always is)
    assert "rank 0/2 needs an MI355X" in r.stderr or "rank 1/2 needs an MI355X" in r.stderr
    assert "launching 2 ranks" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]  # no 1-GPU number under an N-GPU label
