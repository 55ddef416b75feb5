"""CPU: index arithmetic of the binary16 hash stage's gradient tile (csrc/kernels_fp16.h, hash16_phase; round 6).

Entry (t, x) of the tile holds the gradients of window rows t and t + 1 as packed pairs: ((gx[t], gx[t+1]), (gy[t], gy[t+1])) with
gx[t] = L[t+2][x+1] - L[t][x+1], gy[t] = L[t+1][x+2] - L[t+1][x] (computeGTWG_Segment_AVX512FP16_16f, Raisr_AVX512FP16.cpp:138-224, reads
them as central differences of the window).  The kernel used to compute every entry on its own (8 window reads); it now walks down the
columns -- wave w owns pair rows [t0, t0 + cnt), lane = column, three window rows of gx in registers -- and handles the ten halo
columns 64..73 one entry per thread.  This replays both index schemes on integers and compares every entry."""
import numpy as np

LW, GW, GH, LH = 77, 74, 25, 28


def old_entries(L):
    out = {}
    for idx in range(GH * GW):
        ty, tx = divmod(idx, GW)
        c = lambda dy, dx: L[ty + dy, tx + 1 + dx]
        out[(ty, tx)] = (c(2, 0) - c(0, 0), c(3, 0) - c(1, 0), c(1, 1) - c(1, -1), c(2, 1) - c(2, -1))
    return out


def new_entries(L):
    out = {}
    for w in range(4):
        t0, cnt = (0, 7) if w == 0 else (6 * w + 1, 6)
        for lane in range(64):
            col = lambda dy, dx: L[t0 + dy, lane + 1 + dx]
            la, lb = col(0, 0), col(1, 0)
            gxp = gyp = None
            for i in range(8):
                if i <= cnt:
                    lc = col(i + 2, 0)
                    gx = lc - la
                    gy = col(i + 1, 1) - col(i + 1, -1)
                    if i > 0:
                        key = (t0 + i - 1, lane)
                        assert key not in out
                        out[key] = (gxp, gx, gyp, gy)
                    gxp, gyp, la, lb = gx, gy, lb, lc
    for tid in range(256):
        if tid < GH * (GW - 64):
            ty, r = divmod(tid, GW - 64)
            tx = 64 + r
            c = lambda dy, dx: L[ty + dy, tx + 1 + dx]
            assert (ty, tx) not in out
            out[(ty, tx)] = (c(2, 0) - c(0, 0), c(3, 0) - c(1, 0), c(1, 1) - c(1, -1), c(2, 1) - c(2, -1))
    return out


def test_walking_the_columns_fills_every_entry_with_the_same_differences():
    rng = np.random.default_rng(3)
    L = rng.integers(0, 1024, (LH, LW)).astype(np.int64)
    a, b = old_entries(L), new_entries(L)
    assert set(a) == set(b) and len(a) == GH * GW
    assert all(a[k] == b[k] for k in a)


def test_the_walk_stays_inside_the_staged_window():
    for w in range(4):
        t0, cnt = (0, 7) if w == 0 else (6 * w + 1, 6)
        assert t0 + cnt + 2 <= LH - 1                      # last window row read for gx
    assert sum(7 if w == 0 else 6 for w in range(4)) == GH
