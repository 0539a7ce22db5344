"""The index arithmetic of the STAGED GEMM epilogue (videovanish_amd/csrc/vv_gemm_epilogue.h, round 5) restated on the CPU: a strip of 16 rows x W columns is written to a
wave-private LDS tile in the accumulator layout and read back row-major.  The read-back must visit every element exactly once, never let a lane's vector straddle two rows, and
hand each wave instruction whole contiguous row segments -- for the fp32 form (float4 per lane) and for the opt-in h16 form (8 columns per lane, the last instruction half empty
when 16 W is not a multiple of 512)."""
import numpy as np
import pytest


@pytest.mark.parametrize("NT", [4, 5])
def test_staged_fp32_readback_covers_the_strip_once(NT):
    W, pitch = NT * 16, NT * 16 + 4
    tile = np.full(16 * pitch, -1, np.int64)
    for lane in range(64):                                   # accumulator layout: lane (lr, lq) owns row lr, columns 16 j + 4 lq .. + 3
        lr, lq = lane & 15, lane >> 4
        for j in range(NT):
            for e in range(4):
                tile[lr * pitch + j * 16 + 4 * lq + e] = lr * W + j * 16 + 4 * lq + e
    seen = np.zeros(16 * W, np.int64)
    for q in range(NT):
        rows = set()
        for lane in range(64):
            idx = (q * 64 + lane) * 4
            rr, cc = idx // W, idx % W
            assert cc + 3 < W                                # a float4 never straddles two rows
            got = tile[rr * pitch + cc: rr * pitch + cc + 4]
            assert list(got) == [rr * W + cc + e for e in range(4)]
            seen[got] += 1
            rows.add(rr)
        assert len(rows) <= 256 // W + 2                    # an instruction covers ~256 / W consecutive rows, each as one contiguous segment
    assert (seen == 1).all()


@pytest.mark.parametrize("NT", [4, 5])
def test_staged_h16_readback_covers_the_strip_once(NT):
    W = NT * 16
    nq = (16 * W + 511) // 512
    seen = np.zeros(16 * W, np.int64)
    for q in range(nq):
        for lane in range(64):
            idx = (q * 64 + lane) * 8
            if idx >= 16 * W:                                # the half-empty last instruction of the 80-column tile
                continue
            rr, cc = idx // W, idx % W
            assert cc + 7 < W and cc % 8 == 0                # 8 columns of one row: a 16-byte h16 store, 32 bytes of fp32 residual
            seen[rr * W + cc: rr * W + cc + 8] += 1
    assert (seen == 1).all()
    assert nq == (3 if NT == 5 else 2)
