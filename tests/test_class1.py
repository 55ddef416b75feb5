"""CPU: the class-1 rule of the certified hash stage (exactly one-dimensional windows; csrc/kernels_hash_certify.h class1_coherence /
class1_buckets, docs/CERTIFY.md s9) against the oracle's hash (oracle/raisr_oracle.c hash_pixel <- Raisr_AVX512.cpp:175-258, Raisr_AVX256.cpp:393-472).

A window in which every gy (or every gx) is 0 has the reference tensor (a, +-0, 0) [(0, +-0, d)] EXACTLY.  The rule says its bucket is
   angle index  = the zero tensor's angle index (b == 0: xx = 1, a constant angle),
   strength     = by the bound of approx_hash (checked elsewhere),
   coherence    = 2 in the AVX2 flavour;  in the AVX-512 flavour 0 when L2 = fl(a/2 - sqrt14(fl(a a)/4)) < 0 and 2 otherwise,
                  a function of the mantissa of a alone, tabulated per bucket of 128 consecutive floats.
This file enumerates ALL 2^23 mantissas against the oracle, checks the scale invariance over the binades content can reach, replays the
kernel's bucket logic on random boxes, and keeps the constants it relies on in step with the sources."""
import os
import re

import numpy as np
import pytest

from common import folder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = [("filters_2x/filters_highres", 8, 1), ("filters_2x/filters_highres", 8, 2), ("filters_2x/filters_highres", 10, 1),
          ("filters_2x/filters_lowres", 8, 1), ("filters_2x/filters_denoise", 8, 1), ("filters_2x/filters_denoise", 10, 2),
          ("filters_1.5x/filters_denoise", 8, 1), ("filters_1.5x/filters_highres", 8, 1)]


def _pass(fold, bits, pno, asm=2):
    import oracle_py as O
    return O.make_pass(O.Model(folder(fold), bits, pno), bits, False, asm)


def _hash(a, P, avx2, which=0, b=0.0):
    import oracle_py as O
    abd = np.zeros((a.size, 3), np.float32)
    abd[:, 2 * which] = a
    abd[:, 1] = b
    return O.hash_array(abd, P, avx2).astype(np.int32)


def _mantissas(step=1, exp=127):
    m = np.arange(0, 1 << 23, step, dtype=np.uint32)
    return (m | np.uint32(exp << 23)).view(np.float32)


def reference_table():
    """c1tab as k_build_c1tab defines it, from the ORACLE's hash of (a, 0, 0), a in [1, 2): bit 0 = some a of the bucket has coherence index 0
    (L2 < 0 -> NaN), bit 1 = some has index 2."""
    P = _pass(*MODELS[0])
    ci = _hash(_mantissas(), P, False) % 3
    assert set(np.unique(ci)) <= {0, 2}
    flags = np.where(ci == 0, 1, 2).astype(np.uint8).reshape(65536, 128)
    return np.bitwise_or.reduce(flags, axis=1), ci


def test_every_mantissa_every_shipped_model():
    import oracle_py as O
    tab, ci_ref = reference_table()
    assert 0.25 < (ci_ref == 0).mean() < 0.5                       # ~38 % of the mantissas have L2 < 0: neither sign is rare
    for fold, bits, pno in MODELS:
        P = _pass(fold, bits, pno)
        assert max(P.qcoh[0], P.qcoh[1]) < 0.96                    # the rule's precondition (device_abi.hip make_pass: c1_ok)
        step = 1 if (fold, bits, pno) == MODELS[0] else 5
        a = _mantissas(step)
        zero = O.hash_array(np.zeros((1, 3), np.float32), P, False)[0], O.hash_array(np.zeros((1, 3), np.float32), P, True)[0]
        for which in (0, 1):                                       # (a, 0, 0) and (0, 0, d)
            h = _hash(a, P, False, which)
            assert np.all(h // 9 == zero[0] // 9), (fold, bits, pno)
            assert np.array_equal(h % 3, ci_ref[::step]), (fold, bits, pno)       # the coherence index does not depend on the model
            hl = _hash(a[::3], P, True, which)
            assert np.all(hl // 9 == zero[1] // 9) and np.all(hl % 3 == 2), (fold, bits, pno)
        # b = -0 behaves as +0 (sums of signed zero products)
        assert np.array_equal(_hash(a[::11], P, False, 0, b=-0.0), _hash(a[::11], P, False, 0))


def test_scale_invariance_over_the_binades_content_reaches():
    """a = m 2^e: the coherence index is a function of m alone from the smallest non-zero tensor entry (one 16-bit LSB: 4.5e-15) to the
    largest (~1) -- and beyond, down to the guard of class1_coherence (1e-17) and up to 1e17."""
    _, ci_ref = reference_table()
    P = _pass(*MODELS[0])
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(1 << 23, 40000, replace=False)).astype(np.uint32)
    for e in list(range(127 - 57, 127 + 3)) + [127 + 20, 127 + 56]:
        a = (pick | np.uint32(e << 23)).view(np.float32)
        if a.min() < 1e-17 or a.max() > 1e17:
            continue
        for avx2 in (False, True):
            h = _hash(a, P, avx2)
            assert np.array_equal(h % 3, 2 * np.ones_like(h) if avx2 else ci_ref[pick]), (e, avx2)


def _constants():
    hdr = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "kernels_hash_certify.h")).read()
    hip = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "device_abi.hip")).read()
    assert re.search(r"S\.c1e = \(float\)\(1\.05 \* eps\);", hip), "the box half-width of class 1 moved: this test must follow the code"
    assert "il = (lo >> 7) & (kC1Buckets - 1u), ih = (hi >> 7) & (kC1Buckets - 1u)" in hdr and "constexpr unsigned kC1Buckets = 65536u;" in hdr
    assert "0x3f800000u | (i << 7) | k" in hdr and "k < 128u" in hdr
    m = re.search(r"const double eps = ([0-9.]+) \* \(eps_w \+ ([0-9.]+) \* u\);", hip)
    return float(m.group(1)), float(m.group(2))


def test_bucket_logic_never_certifies_a_wrong_index():
    """Replay of class1_coherence: box [a'(1 - 1.05 eps), a'(1 + 1.05 eps)] in fp32, buckets il..ih (at most three, cyclic), certified when
    they carry one and the same single bit.  For random a' and EVERY float a with |a - a'| <= eps a' (the tensor bound of CERTIFY s1):
    a certified index equals the oracle's.  And the share that is certified stays useful."""
    f = np.float32
    k105, k48 = _constants()
    eps_w = 2.4e-6                                                  # >= the measured 2.33e-6 .. 2.35e-6 (tests/test_certify_bounds.py)
    eps = k105 * (eps_w + k48 * 2.0 ** -24)
    tab, ci_ref = reference_table()
    rng = np.random.default_rng(11)
    n = 60000
    bits = (rng.integers(0, 1 << 23, n).astype(np.uint32) | (rng.integers(127 - 48, 127 + 1, n).astype(np.uint32) << 23))
    a1 = bits.view(np.float32)
    e = f(1.05 * eps) * a1
    lo = (a1 - e).astype(f).view(np.uint32)
    hi = (a1 + e).astype(f).view(np.uint32)
    il, ih = (lo >> 7) & 0xFFFF, (hi >> 7) & 0xFFFF
    span = (ih - il) & 0xFFFF
    assert span.max() <= 2                                           # the kernel refuses longer spans; they do not occur
    im = np.where(span >= 2, (il + 1) & 0xFFFF, il)
    t = tab[il] | tab[im] | tab[ih]
    cert = (span <= 2) & ((t == 1) | (t == 2))
    ci_pred = np.where(t == 1, 0, 2)
    assert 0.6 < cert.mean() < 0.85, cert.mean()
    # every float of the box of the certified ones: walk the box in steps of one ulp on both sides (<= ~95 ulps each way, ulps of the finer binade when the box straddles a power of two)
    a64 = a1.astype(np.float64)
    for k in range(-125, 126):
        cand = (bits.astype(np.int64) + k).astype(np.uint32)         # neighbouring floats (crossing a power of two keeps the order)
        av = cand.view(np.float32).astype(np.float64)
        inside = np.abs(av - a64) <= eps * a64
        sel = cert & inside
        assert np.array_equal(ci_ref[cand[sel] & 0x7FFFFF], ci_pred[sel]), k
    for k in (-126, 126):                                            # ... and the walk covered the whole box
        assert not (np.abs((bits.astype(np.int64) + k).astype(np.uint32).view(np.float32).astype(np.float64) - a64) <= eps * a64).any()


def test_coherence_of_a_window_with_nonnegative_l2_clears_every_threshold():
    """L2 >= 0 -> index 2 needs coh >= both thresholds: the smallest coherence over all mantissas with L2 >= 0 (oracle arithmetic restated in
    numpy on the oracle's own sqrt14) is ~0.986 / 0.965, far above every shipped threshold (<= 0.474) and above the c1_ok limits 0.98 / 0.96."""
    import ctypes
    import oracle_py as O
    L = O.lib()
    L.ora_x86_rcp14.restype = ctypes.c_float; L.ora_x86_rsqrt14.restype = ctypes.c_float
    L.ora_x86_rcp14.argtypes = [ctypes.c_float]; L.ora_x86_rsqrt14.argtypes = [ctypes.c_float]
    f = np.float32
    worst = 1.0
    rng = np.random.default_rng(5)
    for m in rng.choice(1 << 23, 4000, replace=False):
        a = (np.uint32(m) | np.uint32(127 << 23)).view(f)
        rad = f(f(a * a) / f(4))
        s = f(L.ora_x86_rcp14(L.ora_x86_rsqrt14(float(rad))))
        l1, l2 = f(f(a / f(2)) + s), f(f(a / f(2)) - s)
        if l2 < 0:
            continue
        s1 = f(L.ora_x86_rcp14(L.ora_x86_rsqrt14(float(l1))))
        s2 = f(L.ora_x86_rcp14(L.ora_x86_rsqrt14(float(l2)))) if l2 > 0 else f(0)
        coh = f(f(s1 - s2) / f(f(s1 + s2) + f(1e-17)))
        worst = min(worst, float(coh))
    assert worst > 0.98, worst
