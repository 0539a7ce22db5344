#!/usr/bin/env python3
"""LDS bank-conflict model used to pick the K / V tile row pitches of vv_attn.hip / vv_attn32.hip (MI355X_MICROARCH.md, LDS section):
ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31},{32-35,44-47,52-59},{36-43,48-51,60-63},
ds_read_b64(_tr_b16) in two 32-lane halves; bank = (addr/4) % 64; N distinct addresses on a bank in one group = N cycles."""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
G64 = [list(range(32)), list(range(32, 64))]


def cycles(addrs, groups, width):
    tot = 0
    for g in groups:
        bank = {}
        for l in g:
            for k in range(width // 4):
                bank.setdefault((addrs[l] // 4 + k) % 64, set()).add(addrs[l] // 4 + k)
        tot += max(len(v) for v in bank.values())
    return tot


def k_read(pitch, s):      # A operand of S^T = K Q^T: lane (li, lg) reads row li, chunk 4s+lg
    return cycles([(l & 15) * pitch + (s * 4 + (l >> 4)) * 16 for l in range(64)], G128, 16)


def v_read(pitch, d):      # transposed V read: lane (li, lg) addresses row 4lg + li/4, columns 16d + 4(li%4)
    return cycles([(4 * (l >> 4) + ((l & 15) >> 2)) * pitch + (d * 16 + 4 * (l & 3)) * 2 for l in range(64)], G64, 8)


if __name__ == "__main__":
    for D in (32, 40, 64, 80, 160, 512):
        DK, DV = (D + 31) // 32 * 32, (D + 15) // 16 * 16
        bk = min((max(k_read(DK * 2 + pad, s) for s in range(DK // 32)), pad) for pad in range(0, 144, 16))
        bv = min((max(v_read(DV * 2 + pad, d) for d in range(DV // 16)), pad) for pad in range(0, 144, 8))
        print(f"D={D}: K pitch pad {bk[1]} -> {bk[0]} cycles (ideal 4); V pitch pad {bv[1]} -> {bv[0]} cycles (ideal 2)")


# ---- round 5: the REAL access set of vv_attn32.hip::attn40q2_kernel per 64-key tile and wave (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
#      19.8 % in the shipped kernel while DESIGN said "no LDS bank conflicts").  Dense 80-byte K / V rows (PR = 80), constant chunk of 1.0 behind the tile.
def attn40q2_access_set(PR=80, TILE=64 * 80):
    """-> list of (name, addrs[64], groups, width bytes).  K fragment reads are ds_read_b128 (lane = (r = lane & 31, h = lane >> 5)), V reads are
    ds_read_b64_tr_b16; the second query block of the wave re-uses every fragment (CSE), so this is the whole per-tile traffic of one wave."""
    acc = []
    for kb in range(2):
        for s in range(3):
            a = []
            for l in range(64):
                r, h = l & 31, l >> 5
                if s < 2:
                    a.append(kb * 32 * PR + r * PR + 16 * h + 32 * s)
                else:
                    a.append(TILE if h else kb * 32 * PR + r * PR + 64)
            acc.append((f"K kb{kb} s{s} (b128)", a, G128, 16))
    for kb in range(2):
        for s2 in range(2):
            g0 = 16 * (2 * kb + s2) * PR
            for half in range(2):
                a = []
                for l in range(64):
                    vq, vpp, vcb, h = (l >> 2) & 3, l & 3, (l >> 4) & 1, l >> 5
                    a.append(g0 + half * 2 * PR + (4 * vq + h) * PR + (16 * vcb + 4 * vpp) * 2)
                acc.append((f"V^T d0..31 kb{kb} s2={s2} {'hi' if half else 'lo'} (b64_tr)", a, G64, 8))
        for half in range(2):
            a = []
            for l in range(64):
                vq, vpp, vcb, h = (l >> 2) & 3, l & 3, (l >> 4) & 1, l >> 5
                a.append(TILE if vpp == 2 else kb * 32 * PR + half * 2 * PR + (16 * vcb + 4 * vq + h) * PR + 64 + 8 * (vpp & 1))
            acc.append((f"V^T d32..47 kb{kb} {'hi' if half else 'lo'} (b64_tr)", a, G64, 8))
    return acc


def attn40q2_report():
    tot = ideal = 0
    rows = []
    for name, a, groups, width in attn40q2_access_set():
        c = cycles(a, groups, width)
        i = len(groups)
        tot += c; ideal += i
        rows.append((name, c, i))
    for name, c, i in rows:
        print(f"  {name:38s} {c:2d} cycles (conflict-free {i})")
    print(f"  per tile and wave: {tot} LDS cycles, {tot - ideal} of them conflicts = {100.0 * (tot - ideal) / tot:.1f} %   "
          f"(measured, profiles/r4_attn40q2_pmc.txt: SQ_LDS_IDX_ACTIVE 60.5, SQ_LDS_BANK_CONFLICT 12.0 per tile and wave = 19.8 %)")
    return tot, tot - ideal


# ---- round 5, second session: the wave-private fp32 strip tile of the STAGED GEMM epilogue (vv_gemm_epilogue.h): ds_write_b128 in the accumulator layout
#      (lane (lr, lq) -> row lr, columns 16 j + 4 lq), ds_read_b128 row-major (flat index 4 (64 q + lane) -> row, column)
def staged_tile_report():
    for W in (80, 64):
        for pad in (0, 4, 8):
            P = W + pad
            wc = [cycles([((l & 15) * P + j * 16 + 4 * (l >> 4)) * 4 for l in range(64)], G128, 16) for j in range(W // 16)]
            rc = [cycles([(((q * 64 + l) * 4 // W) * P + (q * 64 + l) * 4 % W) * 4 for l in range(64)], G128, 16) for q in range(W // 16)]
            print(f"  W = {W}, pitch W + {pad}: write {wc[0]} cycles per instruction, read {max(rc)} (conflict-free: 4 each)")
    print("  (the kernels use W + 4: 2-way on both sides = 80 LDS cycles per 16 x 80 strip against 60 with W + 8 or W + 0; the strip's residual read and store are ~1000x that)")


if __name__ == "__main__":
    print("attn40q2_kernel, per 64-key tile and wave:")
    attn40q2_report()
    print("staged GEMM epilogue tile:")
    staged_tile_report()
