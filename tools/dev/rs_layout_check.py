#!/usr/bin/env python3
"""CPU check of the LDS layout of csrc/aid_gemm_rs.hip: the image the LDS-DMA pieces write (flat 40 / 20 KB slice, 16-B chunks of every
1280-B virtual row XOR-permuted inside their 256-B groups by (virtual row & 15)) against the fragment read addresses
fr[a][t & 3] + 256 (t >> 2) of the 16x16x32 products — every lane reads the chunk it should, and the 16 lanes of every ds_read_b128
lane group (MI355X_MICROARCH.md, LDS table) hit 16 distinct 16-B slots of the bank row.  No GPU needed."""


def frags(K, lane, a):
    i, kg = lane & 15, lane >> 4
    out = []
    for tt in range(4):
        if K == 640:
            out.append((16 * a + i) * 1280 + (((4 * tt + kg) ^ i) << 4))
        else:
            v, u = 8 * a + (i >> 1), i & 1
            out.append(v * 1280 + 512 * u + (256 * u if tt >= 2 else 0) + (((4 * (tt ^ (2 * u)) + kg) ^ v) << 4))
    return out


def check(K, NW):
    SLICE = 32 * K * 2
    nch, PPW = SLICE // 16, SLICE // 1024 // NW
    lds = [-1] * nch
    for wave in range(NW):
        for j in range(PPW):
            for lane in range(64):
                pos = (wave * PPW + j) * 64 + lane
                v, cp = pos // 80, pos % 80
                lds[pos] = v * 80 + (cp ^ (v & 15))
    assert sorted(lds) == list(range(nch))
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    worst = 1
    for a in range(2):
        for t in range(K // 32):
            addr = {}
            for lane in range(64):
                ad = frags(K, lane, a)[t & 3] + 256 * (t >> 2)
                assert ad % 16 == 0 and lds[ad // 16] == (16 * a + (lane & 15)) * (K // 8) + 4 * t + (lane >> 4), (K, a, t, lane)
                addr[lane] = ad
            for g in groups:
                slots = {}
                for lane in g:
                    slots.setdefault((addr[lane] // 16) % 16, []).append(lane)
                worst = max(worst, max(len(x) for x in slots.values()))
    return worst


if __name__ == "__main__":
    for K, NW in ((640, 8), (320, 4)):
        print(f"K = {K}: every fragment read finds its chunk; worst bank conflict {check(K, NW)}-way")
