"""Host-side audit of the fused partial-row plans -- runs WITHOUT a GPU, through the C ABI.

For every 3x3x3 layer of BASELINE config 5 (160x160x96, N=2) and config 2 / 4 (128^3, N=4), in the forward form, the
data-gradient form with GroupNorm-backward sums and the ConvTranspose data-gradient form, `mednet_conv3d_stats_plan` returns the
launch plan made by the launcher's own planning code (kernel, grid, work items, rows per sample).  The test walks that plan
with the row rule the header documents for each kernel and asserts that

  * the set of (sample, row, channel block) slots the kernel WRITES equals the set the reducers READ -- all `rows` rows of
    all channel blocks of all samples (mednet_gn_finalize / mednet_gn_act_bwd_fused read partial[n][0..rows)[0..C)) -- each
    written exactly once;
  * every (brick, channel block) of the output is computed exactly once;
  * a workgroup that accumulates over its items sees non-decreasing sample indices and a constant channel block (its
    running sums are flushed when the sample changes, never re-opened).
The rows the producers of `mednet_head_dgrad_gn` / `mednet_pool2_bwd_gn` / the first-layer kernel write are one per wave of a
grid that is derived from the row count itself; they are covered by the GPU poison test (tests/test_gpu_ops.py).
"""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-mednet_amd"))
from mednet_hip import _lib as L  # noqa: E402

BF16, F16 = L.BF16, L.F16


def plan(n, d, h, w, cin, cout, dtype, gnb, stride, split=False):
    lib = L.lib()
    lib.mednet_set_option(b"assume_cus", 256)  # the MI355X's CU count (the 32 -> 32 specialisation is laid out for it)
    out = (C.c_int * 13)()
    rc = lib.mednet_conv3d_stats_plan(n, d, h, w, cin, cout, dtype, int(gnb) | (2 if split else 0), stride, C.addressof(out))
    lib.mednet_set_option(b"assume_cus", 0)
    assert rc == 0, lib.mednet_last_error().decode()
    keys = ("kind", "grid", "nitems", "ncb", "ntiles", "tps", "accum", "rows", "xcd_chunk", "zslab", "tiles_x", "tiles_y", "tiles_z")
    return dict(zip(keys, out))


def audit(n, p):
    """Walk the plan; returns (written slots as a dict (sample, row, cb) -> count, computed (brick, cb) -> count)."""
    written, computed = {}, {}

    def put(table, key):
        table[key] = table.get(key, 0) + 1

    if p["kind"] in (2, 3, 5, 6):
        # 2 / 3: conv_mfma_kernel (two workgroups per CU); 5 / 6: conv2b_mfma_kernel (round 6: one workgroup per CU, `ncb` counts PAIRS
        # of channel blocks -- a wave writes the sums of its pair's 64 channels into its row) -- the same item numbering and row rules
        grid, ncb = p["grid"], p["ncb"]
        if p["kind"] in (5, 6):
            assert grid == min(256, p["nitems"]) and grid % 8 == 0, p
        for b in range(grid):
            items = []
            i = b
            while i < p["nitems"]:
                lq, cb = (i >> 3) // ncb, (i >> 3) % ncb
                # (conv2b with a brick count that is a multiple of 8: XCD i & 7 owns a contiguous run of xcd_chunk bricks)
                tile = (i & 7) * p["xcd_chunk"] + lq if (p["kind"] in (5, 6) and p["xcd_chunk"]) else lq * 8 + (i & 7)
                if tile >= p["ntiles"]:
                    break  # padding item: the kernel returns (first item) or stops (later ones)
                items.append((tile, cb))
                i += grid
            for tile, cb in items:
                put(computed, (tile, cb))
            if p["kind"] in (2, 5):
                for tile, cb in items:
                    for wv in range(4):
                        put(written, (tile // p["tps"], (tile % p["tps"]) * 4 + wv, cb))
            elif items:  # accumulate mode: one row per wave for EVERY sample, in the workgroup's (constant) channel block
                assert len({cb for _, cb in items}) == 1, f"workgroup {b} changes its channel block"
                samples = [t // p["tps"] for t, _ in items]
                assert samples == sorted(samples), f"workgroup {b} re-opens a sample: {samples}"
                row0 = ((b >> 3) // ncb * 8 + (b & 7)) * 4
                for nn in range(n):
                    for wv in range(4):
                        put(written, (nn, row0 + wv, items[0][1]))
    elif p["kind"] == 4:
        grid = p["grid"]
        assert grid == 256 and p["ncb"] == 1
        for b in range(grid):
            xcd = b & 7
            seq, step, end = (b >> 3, grid >> 3, p["xcd_chunk"]) if p["xcd_chunk"] else (b, grid, p["ntiles"])
            samples = []
            q = seq
            while q < end:
                if p["zslab"]:
                    tx, qd = q % p["tiles_x"], q // p["tiles_x"]
                    ty, zz = qd // p["zslab"], xcd * p["zslab"] + qd % p["zslab"]
                    nn, tz = zz // p["tiles_z"], zz % p["tiles_z"]
                else:
                    tt = xcd * p["xcd_chunk"] + q if p["xcd_chunk"] else q
                    tx, tt = tt % p["tiles_x"], tt // p["tiles_x"]
                    ty, tt = tt % p["tiles_y"], tt // p["tiles_y"]
                    tz, nn = tt % p["tiles_z"], tt // p["tiles_z"]
                assert 0 <= tx < p["tiles_x"] and 0 <= ty < p["tiles_y"] and 0 <= tz < p["tiles_z"] and 0 <= nn < n, (b, q)
                put(computed, (((nn * p["tiles_z"] + tz) * p["tiles_y"] + ty) * p["tiles_x"] + tx, 0))
                samples.append(nn)
                q += step
            assert samples == sorted(samples), f"workgroup {b} re-opens a sample"
            for nn in range(n):  # (workgroups without bricks still owe their zero rows)
                for wv in range(4):
                    put(written, (nn, b * 4 + wv, 0))
    elif p["kind"] == 7:
        # convt_dgrad32_mfma_kernel (round 6): one workgroup per CU, wave = (channel block, z-plane of the 2 x 4 x 16 brick); a wave keeps
        # its sums over the workgroup's bricks of a sample: row 2 * workgroup + z-plane, the two waves of a plane write its two halves
        grid = p["grid"]
        assert grid == 256 and p["ncb"] == 2 and p["rows"] == 2 * grid and p["ntiles"] >= grid
        for b in range(grid):
            seq, step, end = (b >> 3, grid >> 3, p["xcd_chunk"]) if p["xcd_chunk"] else (b, grid, p["ntiles"])
            samples = []
            for q in range(seq, end, step):
                tile = (b & 7) * p["xcd_chunk"] + q if p["xcd_chunk"] else q
                assert 0 <= tile < p["ntiles"]
                for cb in range(2):
                    put(computed, (tile, cb))
                samples.append(tile // p["tps"])
            assert samples == sorted(samples), f"workgroup {b} re-opens a sample"
            for nn in range(n):
                for nt in range(2):
                    for cb in range(2):
                        put(written, (nn, 2 * b + nt, cb))
    else:
        raise AssertionError(f"unknown plan kind {p['kind']}")
    return written, computed


def check(n, d, h, w, cin, cout, dtype, gnb, stride, split=False):
    p = plan(n, d, h, w, cin, cout, dtype, gnb, stride, split)
    written, computed = audit(n, p)
    what = f"n={n} {d}x{h}x{w} {cin}->{cout} gnb={gnb} stride={stride} split={split} plan={p}"
    want_rows = {(nn, r, cb) for nn in range(n) for r in range(p["rows"]) for cb in range(p["ncb"])}
    missing = want_rows - set(written)
    extra = set(written) - want_rows
    assert not missing, f"{what}: {len(missing)} partial rows are read by the reducer but never written, e.g. {sorted(missing)[:4]}"
    assert not extra, f"{what}: rows written outside the buffer the reducer reads, e.g. {sorted(extra)[:4]}"
    assert all(c == 1 for c in written.values()), f"{what}: rows written more than once"
    want_out = {(t, cb) for t in range(p["ntiles"]) for cb in range(p["ncb"])}
    assert set(computed) == want_out and all(c == 1 for c in computed.values()), f"{what}: output bricks not covered exactly once"
    return p


def unet_layers(f_maps, size, n):
    """(n, d, h, w, cin, cout) of every 3x3x3 Conv3d of ResidualUNet3D(f_maps) except the 1-channel first layer, and (n, d, h, w,
    cin, cout) of every ConvTranspose3d at its LOW-resolution grid (model.py:140-214, components.py:136-180,259-264)."""
    convs, convts = [], []
    d, h, w = size
    for lvl, f in enumerate(f_maps):
        if lvl:
            d, h, w = d // 2, h // 2, w // 2
            convs.append((n, d, h, w, f_maps[lvl - 1], f))  # conv1 of the encoder's ExtResNetBlock
        convs.append((n, d, h, w, f, f))  # conv2 / conv3 (and the decoder block's three convs at this level)
    d, h, w = size
    dims = [(d >> k, h >> k, w >> k) for k in range(len(f_maps))]
    for lvl in range(len(f_maps) - 1, 0, -1):
        convts.append((n,) + dims[lvl] + (f_maps[lvl], f_maps[lvl - 1]))
    return convs, convts


CFG = {"cfg5": ([64, 128, 256, 512, 1024], (160, 160, 96), 2), "cfg2": ([32, 64, 128, 256], (128, 128, 128), 4),
       "cfg2_n1": ([32, 64, 128, 256], (128, 128, 128), 1), "odd": ([32, 64, 96], (36, 44, 20), 3)}


# which kernels the layers of the two benchmarked configurations go to: conv_mfma_kernel per-brick rows (2) / accumulate mode (3), the
# 32 -> 32 specialisation (4), the two-block kernel per-brick rows (5) / accumulate mode (6), the ConvTranspose3d 64 -> 32 data gradient (7)
EXPECTED_KINDS = {"cfg5": {2, 3, 5, 6}, "cfg2": {2, 3, 4, 6, 7}}


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("cfg", sorted(CFG))
def test_every_partial_row_is_written_exactly_once(cfg, dtype):
    f_maps, size, n = CFG[cfg]
    convs, convts = unet_layers(f_maps, size, n)
    kinds = set()
    for (nn, d, h, w, cin, cout) in convs:
        kinds.add(check(nn, d, h, w, cin, cout, dtype, False, 1)["kind"])  # forward + GroupNorm statistics
        kinds.add(check(nn, d, h, w, cout, cin, dtype, True, 1)["kind"])  # data gradient + GroupNorm-backward sums
    for (nn, d, h, w, cin, cout) in convts:
        kinds.add(check(nn, d, h, w, cin, cout, dtype, True, 2)["kind"])  # ConvTranspose data gradient + GroupNorm-3 sums
    if cfg == "cfg5":
        assert kinds == EXPECTED_KINDS["cfg5"], kinds
    if cfg == "cfg2":
        assert kinds == EXPECTED_KINDS["cfg2"], kinds


@pytest.mark.parametrize("shape", [(9, 11, 21), (4, 8, 16), (130, 70, 34), (64, 64, 64), (5, 300, 17)])
@pytest.mark.parametrize("n", [1, 2, 5])
@pytest.mark.parametrize("chan", [(16, 16), (32, 32), (32, 64), (96, 32), (48, 80)])
def test_ragged_shapes(shape, n, chan):
    """Brick counts that are not multiples of 8 (padding items), 16-channel half blocks, channel block counts that do not
    divide 64 (no accumulate mode), the 32 -> 32 specialisation with and without the z-slab walk."""
    d, h, w = shape
    for gnb in (False, True):
        check(n, d, h, w, chan[0], chan[1], BF16, gnb, 1)
    if chan[0] % 32 == 0 and chan[1] % 32 == 0:
        check(n, d, h, w, chan[0], chan[1], BF16, True, 2)


# ---- weight-gradient plans (round 5: wgrad_mfma4_kernel's z-columns, wgrad_mfma2_kernel<8>'s narrow bricks) -----------------------
def wgrad_plan(n, d, h, w, cin, cout, dtype, workgroups):
    lib = L.lib()
    out = (C.c_int * 10)()
    rc = lib.mednet_conv3d_wgrad_plan(n, d, h, w, cin, cout, dtype, workgroups, C.addressof(out))
    assert rc == 0, lib.mednet_last_error().decode()
    keys = ("kind", "grid", "pairs", "splits", "nitems", "tiles_x", "tiles_y", "tz_or_slabs", "zs_or_tx", "xcd_remap")
    return dict(zip(keys, out))


def audit_wgrad(n, d, h, w, p):
    """Every voxel of every sample must be contracted exactly once per channel-block pair: walk the item lists of a pair's
    workgroups with the kernels' own rules (first item, stride, XCD remap) and mark the voxels of each item."""
    import numpy as np
    seen = np.zeros((n, d, h, w), dtype=np.int32)
    splits, nitems = p["splits"], p["nitems"]
    assert p["grid"] == p["pairs"] * splits and 1 <= splits <= max(nitems, 1)
    for split in range(splits):
        first = (split & 7) * (splits >> 3) + (split >> 3) if p["xcd_remap"] else split
        if p["xcd_remap"]:
            assert splits % 8 == 0
        for item in range(first, nitems, splits):
            t = item
            tx, t = t % p["tiles_x"], t // p["tiles_x"]
            ty, t = t % p["tiles_y"], t // p["tiles_y"]
            tz, nn = t % p["tz_or_slabs"], t // p["tz_or_slabs"]
            assert nn < n
            if p["kind"] == 4:
                zs = p["zs_or_tx"]
                z0, z1, y0, x0, ey, ex = tz * zs, min(d, (tz + 1) * zs), ty * 8, tx * 16, 8, 16
            else:
                z0, z1, y0, x0, ey, ex = tz * 4, min(d, tz * 4 + 4), ty * 8, tx * p["zs_or_tx"], 8, p["zs_or_tx"]
            seen[nn, z0:z1, y0:min(h, y0 + ey), x0:min(w, x0 + ex)] += 1
    assert seen.min() == 1 and seen.max() == 1, (p, int(seen.min()), int(seen.max()))


@pytest.mark.parametrize("workgroups", [0, 128, 24])
@pytest.mark.parametrize("cfg", sorted(CFG))
def test_weight_gradient_plans_cover_every_voxel_once(cfg, workgroups):
    f_maps, size, n = CFG[cfg]
    convs, _ = unet_layers(f_maps, size, n)
    kinds = set()
    for (nn, d, h, w, cin, cout) in convs:
        p = wgrad_plan(nn, d, h, w, cin, cout, BF16, workgroups)
        audit_wgrad(nn, d, h, w, p)
        kinds.add((p["kind"], p["zs_or_tx"] if p["kind"] == 2 else 0))
        if p["kind"] == 4:  # slabs of at most 32 planes that cover the depth
            assert p["zs_or_tx"] <= 32 and p["zs_or_tx"] * p["tz_or_slabs"] >= d > p["zs_or_tx"] * (p["tz_or_slabs"] - 1)
    if cfg == "cfg5":  # levels 0 - 2 on the co-resident kernel, 20 x 20 x 12 on 16-wide bricks, 10 x 10 x 6 on 8-wide ones
        assert kinds == {(4, 0), (2, 16), (2, 8)}, kinds
    if cfg == "cfg2":
        assert kinds == {(4, 0)}, kinds


@pytest.mark.parametrize("shape", [(9, 11, 21), (4, 8, 16), (10, 10, 6), (7, 9, 40), (33, 8, 16), (64, 20, 17), (5, 300, 7)])
@pytest.mark.parametrize("chan", [(16, 16), (32, 64), (96, 32), (256, 512)])
def test_weight_gradient_plans_ragged(shape, chan):
    d, h, w = shape
    for n in (1, 3):
        for wgs in (0, 128, 1000):
            audit_wgrad(n, d, h, w, wgrad_plan(n, d, h, w, chan[0], chan[1], F16, wgs))


def test_narrow_brick_plans():
    """8-wide bricks (FwdTile<3>) for volumes up to 8 voxels wide: the forward / data-gradient plans of config 5's deepest level."""
    for gnb in (False, True):
        p = check(2, 10, 10, 6, 1024, 1024, BF16, gnb, 1)
        assert (p["tiles_x"], p["tiles_y"], p["tiles_z"]) == (1, 2, 3) and p["kind"] == 2
        check(3, 7, 13, 8, 64, 96, F16, gnb, 1)
        check(1, 4, 4, 4, 256, 256, BF16, gnb, 1)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg2_n1", "cfg5", "odd"])
def test_split_weight_plans(cfg):
    """fp16x2 (split weights): every stride-1 layer whose output channels come in whole blocks goes to conv2b's (high, low) form --
    `ncb` counts channel BLOCKS there, the grid shrinks below 256 for small layers -- and the rest to the general kernel with the low
    image as extra K chunks (same items, same rows); the ConvTranspose data gradient keeps its plan.  Same coverage rules."""
    f_maps, size, n = CFG[cfg]
    convs, convts = unet_layers(f_maps, size, n)
    kinds = set()
    for (nn, d, h, w, cin, cout) in convs:
        kinds.add(check(nn, d, h, w, cin, cout, F16, False, 1, split=True)["kind"])
        kinds.add(check(nn, d, h, w, cout, cin, F16, True, 1, split=True)["kind"])
    for (nn, d, h, w, cin, cout) in convts:
        kinds.add(check(nn, d, h, w, cin, cout, F16, True, 2, split=True)["kind"])
    assert 4 not in kinds, kinds  # (the 32 -> 32 specialisation keeps one weight image in registers: not in this mode)
