"""CPU: the lane program of the symmetric filter stage (csrc/kernels_filter.h, filter_phase<.., SYM>) replayed in numpy
against the reference's plain 16-lane chains (Raisr_AVX512.cpp:134-149, sumitup_ps_512 :37-44).

For a palindromic bank row (f[k] == f[120 - k]) each lane loads four coefficients instead of eight, runs taps ch = 0..3 of its
own chain, hands the accumulator to lane (8 - l) & 15 and continues there with the partner chain's taps ch = 4..7 on the same
four registers.  The claim the kernel relies on: all 16 chains end with the reference's bits, except that a chain whose
eight products are all -0 may end with -0 where the reference's padding step made it +0 (lanes >= 9 run that step as
fma(+0, c3, acc) in the middle of the chain); the pixel's sum v then has the reference's bits unless it is a zero, which the
accept test (v > clamp_lo >= 0) rejects whatever its sign.
Also: which of the shipped banks qualify (the facts DESIGN.md quotes)."""
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def fma(a, b, c):
    # a*b is exact in binary64 for binary32 operands; the sum is rounded to binary64 then binary32.  Double rounding could differ
    # from a true fma in rare ties, but both sides of the comparison use this same function on the same operand triples.
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def tree(a):
    r8 = [f32(a[i] + a[(i + 8) % 16]) for i in range(16)]
    r4 = [f32(r8[i] + r8[(i + 4) % 16]) for i in range(16)]
    r2 = [f32(r4[i] + r4[(i + 2) % 16]) for i in range(16)]
    return [f32(r2[i] + r2[(i + 1) % 16]) for i in range(16)]


def reference_lanes(p, f):
    acc = [f32(p[l] * f[l]) for l in range(16)]
    for ch in range(1, 8):
        acc = [fma(p[16 * ch + l], f[16 * ch + l], acc[l]) for l in range(16)]
    return acc, tree(acc)


def symmetric_lanes(p, f):
    cf = [[f[16 * c + l] for c in range(4)] for l in range(16)]          # the only coefficients a lane loads
    a = [f32(p[l] * cf[l][0]) for l in range(16)]
    for c in range(1, 4):
        a = [fma(p[16 * c + l], cf[l][c], a[l]) for l in range(16)]
    a = [a[(8 - q) % 16] for q in range(16)]                               # partner_xchg
    for q in range(16):
        l2 = (8 - q) % 16
        if q <= 8:
            taps = [16 * (4 + j) + l2 for j in range(4)]
            g = [cf[q][3], cf[q][2], cf[q][1], cf[q][0]]
        else:                                                              # padding step first (the lane reads +0 for the pixel), then taps ch = 4, 5, 6
            taps = [None] + [16 * (3 + j) + l2 for j in range(1, 4)]
            g = [cf[q][3], cf[q][2], cf[q][1], cf[q][0]]
        for j in range(4):
            x = f32(0) if taps[j] is None else p[taps[j]]
            a[q] = fma(x, g[j], a[q])
    # lane q holds chain (8 - q) & 15: undo the permutation for the chain-by-chain comparison (the tree is invariant under it:
    # test_partner_map_is_a_tree_automorphism)
    return [a[(8 - l) % 16] for l in range(16)], tree(a)


def palindrome_row(rng, negatives=False):
    h = (rng.standard_normal(61) * 0.1).astype(f32)
    if negatives:
        h = -np.abs(h)
    f = np.zeros(128, f32)
    f[:61] = h
    f[60:121] = h[::-1]
    return f


@pytest.mark.parametrize("seed", range(4))
def test_symmetric_lane_program_gives_the_reference_bits(seed):
    rng = np.random.default_rng(seed)
    zero_sign_cases = [0]
    for trial in range(400):
        f = palindrome_row(rng, negatives=trial % 7 == 0)
        p = np.zeros(128, f32)
        p[:121] = rng.integers(0, 1024, 121).astype(f32)
        if trial % 5 == 0:
            p[:121] *= rng.random(121) < 0.1                               # mostly zero patches: signed-zero chains
        if trial % 11 == 0:
            p[:] = 0
        (wc, wv), (gc, gv) = reference_lanes(p, f), symmetric_lanes(p, f)
        for l in range(16):
            same = wc[l].view(np.uint32) == gc[l].view(np.uint32)
            assert same or (l >= 9 and wc[l] == 0 and gc[l] == 0), (trial, l)      # only a padded chain, only the sign of a zero
        for l in range(16):                                                          # every lane of the tree carries v
            assert wv[l].view(np.uint32) == gv[l].view(np.uint32) or (wv[l] == 0 and gv[l] == 0), (trial, l)
        zero_sign_cases[0] += int(any(wc[l].view(np.uint32) != gc[l].view(np.uint32) for l in range(16)))
    if seed == 0:
        assert zero_sign_cases[0] > 0          # the sweep does reach the one case that differs (all products -0, c3 < 0)


def lane_major(f):
    """k_lane_major_bank: float (ch >> 2) * 64 + 4 l + (ch & 3) of the row holds tap 16 ch + l."""
    out = np.zeros(128, f32)
    for e in range(128):
        ch, l = e >> 4, e & 15
        out[(ch >> 2) * 64 + l * 4 + (ch & 3)] = f[e]
    return out


def redo_lanes(p, f):
    """The second run of a step that holds a pixel of a non-palindromic row (round 6): the symmetric lane program, but what a lane
    multiplies after the hand-over comes from a second 16-byte load -- lane l2 = (8 - q) & 15's run of taps 64 + l2 ... 112 + l2 in
    the lane-major row, 256 B + 16 l2 -- instead of the mirrored registers; lanes >= 9 load from 4 bytes earlier, so that the padding
    step (window value +0) meets the float in front of the run (tap 111 + l2: tap 120 or a padding zero) and taps 64 + l2, 80 + l2,
    96 + l2 the same registers as in lanes <= 8.  The same program with both loads in EVERY step is the 32-byte form of a tile row (MIX)."""
    lm = lane_major(f)
    a = [None] * 16
    fa = [[lm[4 * l + c] for c in range(4)] for l in range(16)]
    for l in range(16):
        acc = f32(p[l] * fa[l][0])
        for c in range(1, 4):
            acc = fma(p[16 * c + l], fa[l][c], acc)
        a[l] = acc
    a = [a[(8 - q) % 16] for q in range(16)]                               # partner_xchg
    for q in range(16):
        l2 = (8 - q) % 16
        poff_floats = 64 + 4 * l2                                          # (256 + 16 l2) bytes from the row's start
        fb = [lm[poff_floats + j] for j in range(4)]
        if q <= 8:
            taps = [16 * (4 + j) + l2 for j in range(4)]
            m = [fb[0], fb[1], fb[2], fb[3]]
        else:                                                              # the lane loads from 4 bytes earlier: the float in front meets the zero block
            taps = [None] + [16 * (3 + j) + l2 for j in range(1, 4)]
            m = [lm[poff_floats - 1], fb[0], fb[1], fb[2]]
            assert np.isfinite(m[0])
        for j in range(4):
            x = f32(0) if taps[j] is None else p[taps[j]]
            a[q] = fma(x, m[j], a[q])
    return [a[(8 - l) % 16] for l in range(16)], tree(a)


def _patch(rng, zeros=0.0):
    p = np.zeros(128, f32)
    p[:121] = rng.integers(0, 1024, 121).astype(f32)
    if zeros:
        p[:121] *= rng.random(121) >= zeros
    return p


@pytest.mark.parametrize("seed", range(3))
def test_redo_lane_program_gives_the_reference_bits_on_any_row(seed):
    rng = np.random.default_rng(100 + seed)
    for trial in range(300):
        kind = trial % 3
        if kind == 0:                                                      # one tap pair off, as in filterbin_2_10
            f = palindrome_row(rng)
            k = int(rng.integers(0, 60))
            f[120 - k] = np.nextafter(f[120 - k], f32(10.0)) if trial % 2 else f32(f[120 - k] + f32(1e-6))
        elif kind == 1:                                                    # nowhere near a palindrome
            f = np.zeros(128, f32)
            f[:121] = (rng.standard_normal(121) * 0.1).astype(f32)
        else:                                                              # a palindrome: the second run reproduces the first
            f = palindrome_row(rng, negatives=trial % 2 == 0)
        p = _patch(rng, zeros=0.9 if trial % 5 == 0 else 0.0)
        (wc, wv), (gc, gv) = reference_lanes(p, f), redo_lanes(p, f)
        for l in range(16):
            assert wc[l].view(np.uint32) == gc[l].view(np.uint32) or (l >= 9 and wc[l] == 0 and gc[l] == 0), (trial, l)
            assert wv[l].view(np.uint32) == gv[l].view(np.uint32) or (wv[l] == 0 and gv[l] == 0), (trial, l)
        if kind == 2:                                                      # ... bit for bit, up to the sign of a zero chain of a padded lane
            (sc, sv) = symmetric_lanes(p, f)                               # (its padding step multiplies +0 by another finite coefficient)
            for l in range(16):
                assert sc[l].view(np.uint32) == gc[l].view(np.uint32) or (l >= 9 and sc[l] == 0 and gc[l] == 0), (trial, l)
                assert sv[l].view(np.uint32) == gv[l].view(np.uint32) or (sv[l] == 0 and gv[l] == 0), (trial, l)


def test_mismatch_in_the_directly_loaded_pairs_costs_nothing():
    """The stage loads taps 0..63 and mirrors taps 64..120 from 56..0: the pairs (57, 63), (58, 62), (59, 61) are both loaded, so a row
    that differs from a palindrome only there needs no second run (scan_bank_symmetry compares k <= 56)."""
    rng = np.random.default_rng(7)
    for trial in range(200):
        f = palindrome_row(rng)
        for k in (57, 58, 59)[: 1 + trial % 3]:
            f[120 - k] = np.nextafter(f[120 - k], f32(-10.0))
        p = _patch(rng)
        (wc, wv), (gc, gv) = reference_lanes(p, f), symmetric_lanes(p, f)
        for l in range(16):
            assert wc[l].view(np.uint32) == gc[l].view(np.uint32) or (l >= 9 and wc[l] == 0 and gc[l] == 0), (trial, l)
            assert wv[l].view(np.uint32) == gv[l].view(np.uint32) or (wv[l] == 0 and gv[l] == 0), (trial, l)
    # and a mismatch at k = 56 does cost: the mirrored tap 64 gets the wrong coefficient
    f = palindrome_row(rng)
    f[64] = f32(f[64] + f32(0.25))
    p = _patch(rng)
    p[64] = f32(100)
    assert reference_lanes(p, f)[1][0] != symmetric_lanes(p, f)[1][0]


def test_rows_the_symmetric_stage_runs_twice_per_shipped_bank():
    need = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "filters_*", "*", "filterbin_*"))):
        a = _bank(path)
        need[os.path.relpath(path, ROOT)] = int((a[:, :57] != a[:, ::-1][:, :57]).any(axis=1).sum())
    assert need["filters_2x/filters_highres/filterbin_2_8"] == 2
    assert need["filters_2x/filters_highres/filterbin_2_10"] == 44       # 50 rows are not palindromes, 6 of them only in taps 57..59 / 61..63
    assert need["filters_2x/filters_highres/filterbin_2_10_2"] == 1


def test_partner_map_is_a_tree_automorphism():
    rng = np.random.default_rng(11)
    for _ in range(50):
        a = [f32(x) for x in rng.standard_normal(16) * 50]
        b = [a[(8 - q) % 16] for q in range(16)]
        assert {x.view(np.uint32) for x in tree(a)} == {x.view(np.uint32) for x in tree(b)}


def test_partner_map_is_the_two_dpp_moves():
    # row_mirror then row_ror:9 (lane p reads lane (p - 9) & 15): result[p] = v[15 - ((p - 9) & 15)] = v[(8 - p) & 15]
    v = list(range(16))
    t = [v[15 - p] for p in range(16)]
    r = [t[(p - 9) % 16] for p in range(16)]
    assert r == [(8 - p) % 16 for p in range(16)]


def _bank(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"fp32"
    n = np.frombuffer(raw[4:16], np.uint32)
    return np.frombuffer(raw[16:], np.uint32).reshape(int(n[0]) * int(n[1]), 121)


def test_which_shipped_banks_are_palindromic():
    counts = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "filters_*", "*", "filterbin_*"))):
        a = _bank(path)
        counts[os.path.relpath(path, ROOT)] = int((a != a[:, ::-1]).any(axis=1).sum())
    # the headline model: 2 of 864 rows are not palindromes (two coefficient pairs off by one ulp each)
    assert counts["filters_2x/filters_highres/filterbin_2_8"] == 2
    assert counts["filters_2x/filters_highres/filterbin_2_10_2"] == 2
    assert counts["filters_2x/filters_highres/filterbin_2_10"] == 50
    assert counts["filters_2x/filters_lowres/filterbin_2_8"] > 400      # not symmetric: the eight-load stage runs


def test_shared_tree_of_sixteen_steps_gives_each_lane_its_step():
    """filter_phase folds the row's 16 steps with ONE tree: after each level two steps are merged on a lane-index bit and the next
    level runs once for both (level 2 pairs lanes inside their half of the row: row_shl:4 on even, row_shr:4 on odd partial sums).
    Lane l must end with the reference's v of step bitrev4(l)."""
    rng = np.random.default_rng(5)
    acc = (rng.standard_normal((16, 16)) * 100).astype(f32)           # acc[s][lane]
    want = [tree(list(acc[s]))[0] for s in range(16)]
    lanes = range(16)
    A = [[f32(acc[s][i] + acc[s][(i + 8) % 16]) for i in lanes] for s in range(16)]
    B = []
    for k in range(8):
        t = [A[2 * k + 1][i] if i & 8 else A[2 * k][i] for i in lanes]
        if k & 1:
            B.append([f32(t[i] + (t[i - 4] if i - 4 >= 0 else f32(0))) for i in lanes])           # row_shr:4: lane i reads lane i - 4
        else:
            B.append([f32(t[i] + (t[i + 4] if i + 4 < 16 else f32(0))) for i in lanes])           # row_shl:4: lane i reads lane i + 4
    C = []
    for m in range(4):
        t = [B[2 * m + 1][i] if i & 4 else B[2 * m][i] for i in lanes]
        C.append([f32(t[i] + t[i ^ 2]) for i in lanes])
    D = []
    for n in range(2):
        t = [C[2 * n + 1][i] if i & 2 else C[2 * n][i] for i in lanes]
        D.append([f32(t[i] + t[i ^ 1]) for i in lanes])
    v = [D[1][i] if i & 1 else D[0][i] for i in lanes]
    for l in lanes:
        sl = ((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3)
        assert v[l].view(np.uint32) == want[sl].view(np.uint32), (l, sl)


def test_lane_major_bank_layout_feeds_each_lane_its_chain():
    """k_lane_major_bank (csrc/kernels_filter.h) permutes every 128-float bank row so that zmm lane l finds taps l, 16 + l, ..., 112 + l
    as two runs of four floats: float (ch >> 2) * 64 + 4 l + (ch & 3) holds tap 16 ch + l.  The filter stage reads run 0 (and run 1,
    256 B further on, in the eight-load stage) with one 16-byte load each."""
    row = np.arange(128, dtype=np.int32)                                   # tap index as the value
    out = np.empty_like(row)
    for e in range(128):                                                   # the kernel's index arithmetic
        ch, l = e >> 4, e & 15
        out[(ch >> 2) * 64 + l * 4 + (ch & 3)] = row[e]
    for l in range(16):
        assert list(out[4 * l:4 * l + 4]) == [16 * c + l for c in range(4)]
        assert list(out[64 + 4 * l:64 + 4 * l + 4]) == [16 * c + l for c in range(4, 8)]
    assert sorted(out) == list(range(128))                                 # a permutation: nothing lost, padding taps 121..127 included
