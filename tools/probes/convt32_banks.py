#!/usr/bin/env python3
"""Bank check of convt_dgrad32_mfma_kernel's LDS image (csrc/convt_dgrad32_mfma.inc), no GPU needed.

A ds_read_b128 is served 16 lanes at a time, in the hardware's lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of each
half-wave (measured in round 6: tools/probes/conv2b_banks.py); a group is conflict-free when its 16 slots of 16 bytes are distinct
modulo 16 (64 banks x 4 bytes).  The image: slot = ((piece * 5 + hz) * 9 + hy) * 34 + (hx & 1) * 17 + (hx >> 1); lane r of an N-tile
reads output column lx(r) of row 2 t + (r >> 4) at halo position (2 nt + kz, 2 row + ky, 2 lx + kx), k-half h = piece 2 kc + h.
Searches the rotation of the second row's lanes that makes every tap of every tile conflict-free."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
ROWP = 34


def slot(piece, hz, hy, hx):
    return ((piece * 5 + hz) * 9 + hy) * ROWP + (hx & 1) * 17 + (hx >> 1)


def worst(rot):
    w = 1
    for nt in range(2):
        for t in range(2):
            for kc in range(2):
                for kz in range(3):
                    for ky in range(3):
                        for kx in range(3):
                            for h in range(2):
                                for grp in GROUPS:
                                    banks = {}
                                    for r in grp:
                                        lx = ((r & 15) - rot * (r >> 4)) & 15
                                        s = slot(2 * kc + h, 2 * nt + kz, 2 * (2 * t + (r >> 4)) + ky, 2 * lx + kx)
                                        banks[s % 16] = banks.get(s % 16, 0) + 1
                                    w = max(w, max(banks.values()))
    return w


if __name__ == "__main__":
    res = {rot: worst(rot) for rot in range(16)}
    print("worst lanes per bank group, by rotation of the second row:", res)
    print("conflict-free rotations:", [r for r, v in res.items() if v == 1])
