"""Debug aids for the determinism audit of the hot path.  Nothing here runs in a normal step.

* `trace(...)`: when a trace is open (`open_trace()`), every op of the path records a bitwise checksum of what it produced
  (activations, GroupNorm partial rows, statistics, gradients) under a running name; two runs of the same step can then be
  compared trace point by trace point and the FIRST diverging tensor names the kernel (tools/probes/determinism_probe.py,
  tests/test_gpu_network.py).
* `poison(...)`: fills the caching allocator's free memory and the per-stream workspaces with a bit pattern (NaN by
  default) so that any kernel reading memory no producer wrote -- an unwritten GroupNorm partial row, a workspace slot -- turns
  the result into NaN deterministically instead of into "whatever the previous step left there".
"""
from __future__ import annotations

import torch

TRACE = None  # list of (name, checksum tensor) while a trace is open


def open_trace():
    global TRACE
    TRACE = []
    return TRACE


def close_trace():
    """-> list of (name, int checksum); synchronises."""
    global TRACE
    t, TRACE = TRACE, None
    if t is None:
        return []
    torch.cuda.synchronize()
    return [(n, int(c.item())) for n, c in t]


def checksum(t: torch.Tensor) -> torch.Tensor:
    """Order-dependent 64-bit checksum of the tensor's BYTES as they lie in memory (NaNs included): sum of the 32-bit
    words plus a position-weighted sum, so that two values changing places do not cancel."""
    if not t.is_contiguous():
        if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
            t = t.permute(0, 2, 3, 4, 1)  # the physical order, a contiguous view
        else:
            t = t.contiguous()
    b = t.reshape(-1).view(torch.uint8)
    pad = (-b.numel()) % 4
    if pad:
        b = torch.cat((b, b.new_zeros(pad)))
    w = b.view(torch.int32).to(torch.int64)
    idx = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
    return w.sum() + ((w * ((idx & 0xFFFF) + 1)).sum() << 1)


def trace(name: str, *tensors):
    """Record checksums of `tensors` (None entries are skipped) under name, name#1, ... -- no-op unless a trace is open."""
    if TRACE is None:
        return
    k = 0
    for t in tensors:
        if t is None or not torch.is_tensor(t) or t.numel() == 0:
            k += 1
            continue
        s = torch.cuda.current_stream(t.device)
        with torch.cuda.stream(s):
            TRACE.append((f"{len(TRACE):04d} {name}#{k}", checksum(t.detach())))
        k += 1


def first_difference(a, b):
    """First trace point whose checksum differs between two closed traces (or whose name does); None when identical."""
    for (na, ca), (nb, cb) in zip(a, b):
        if na != nb:
            return f"trace shapes differ: {na} vs {nb}"
        if ca != cb:
            return na
    if len(a) != len(b):
        return f"trace lengths differ: {len(a)} vs {len(b)}"
    return None


def poison(device="cuda", pattern: int = 0x7FC07FC0, big_gb: float = 48.0, small_mb: int = 1024):
    """Make every byte the caching allocator hands out next carry `pattern` (0x7FC07FC0: a NaN read as fp32, two NaNs read as
    bf16; 0x7FC0 is a NaN in fp16 too) and fill the library's per-stream workspaces with it.  The cached blocks are given
    back to the driver first; then one large block (`big_gb`: what the step's large tensors are carved from) and `small_mb`
    of sub-megabyte blocks (the small pool: GroupNorm partial rows, statistics, coefficients) are allocated, filled and
    freed.  Memory held by live tensors is not touched."""
    from . import _lib
    torch.cuda.synchronize()
    pat = int(torch.tensor([pattern], dtype=torch.int64).to(torch.int32).item())
    for buf in _lib._ws_cache.values():
        v = buf.view(torch.uint8)
        n4 = v.numel() // 4 * 4
        v[:n4].view(torch.int32).fill_(pat)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    held = [torch.empty(int(big_gb * (1 << 30)) // 4, dtype=torch.int32, device=device).fill_(pat)]
    for size_kb in (512, 64, 4):
        n = small_mb * 1024 // 3 // size_kb
        held.extend(torch.empty(size_kb * 256, dtype=torch.int32, device=device).fill_(pat) for _ in range(n))
    torch.cuda.synchronize()
    del held
