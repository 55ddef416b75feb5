"""-m gpu: the device hash functions (the code k_hash runs per pixel) against the oracle's hash_pixel on inputs no
frame produces often enough: every exponent, exact powers of four, rsqrt14 results that are exact powers of two,
zeros, negatives, denormals, infinities and NaNs in each of (a, b, d) -- for the AVX-512 flavour (branch-free fast
path + generic fall-back) and the AVX2 flavour (GetHashValue_AVX512_32f_16Elements Raisr_AVX512.cpp:175-258,
GetHashValue_AVX256_32f_8Elements Raisr_AVX256.cpp:393-472)."""
import numpy as np
import pytest

from common import folder

pytestmark = pytest.mark.gpu

SPECIAL_BITS = np.array([
    0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x007fffff, 0x807fffff, 0x00800000, 0x80800000,
    0x3f800000, 0xbf800000, 0x40000000, 0x40800000, 0x3e800000, 0x3f000000, 0x7f7fffff, 0xff7fffff,
    0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x7f800001, 0x7fffffff, 0x3f7fffff, 0x3fffffff,
    0x407fffff, 0x00ffffff, 0x01000000, 0x2edbe6ff, 0x5d5e0b6b, 0x1e3ce508,
], dtype=np.uint32)


def _triples(seed):
    rng = np.random.default_rng(seed)
    parts = []
    # 1. tensor-like triples at every scale: a, d >= 0, |b| <= sqrt(a d) (+ a little slack so the radicand can go negative)
    n = 1 << 20
    scale = np.exp2(rng.uniform(-60, 60, n)).astype(np.float32)
    a = (rng.random(n, dtype=np.float32) * scale).astype(np.float32)
    d = (rng.random(n, dtype=np.float32) * scale).astype(np.float32)
    b = (np.sqrt(a.astype(np.float64) * d) * rng.uniform(-1.0001, 1.0001, n)).astype(np.float32)
    parts.append(np.stack([a, b, d], 1))
    # 2. realistic magnitudes for 8/10-bit content (weights ~1e-6..1e-5 times squared gradients), incl. isotropic / flat ties
    n = 1 << 20
    a = (rng.integers(0, 255 * 255, n) * rng.uniform(1e-7, 1e-4, n)).astype(np.float32)
    d = np.where(rng.random(n) < 0.2, a, (rng.integers(0, 255 * 255, n) * rng.uniform(1e-7, 1e-4, n))).astype(np.float32)
    b = np.where(rng.random(n) < 0.2, 0.0, np.sqrt(a.astype(np.float64) * d) * rng.uniform(-1, 1, n)).astype(np.float32)
    parts.append(np.stack([a, b, d], 1))
    # 3. radicands / eigenvalues that are exact powers of two and four: a = d = 2^k, b = 2^j
    k = np.arange(-120, 120)
    kk, jj = np.meshgrid(k, k)
    parts.append(np.stack([np.exp2(kk.ravel()), np.exp2(jj.ravel()), np.exp2(kk.ravel())], 1).astype(np.float32))
    parts.append(np.stack([np.exp2(kk.ravel()), np.zeros(kk.size), np.exp2(jj.ravel())], 1).astype(np.float32))
    # 4. every combination of special bit patterns in (a, b, d)
    s = SPECIAL_BITS.view(np.float32)
    g = np.stack(np.meshgrid(s, s, s, indexing="ij"), -1).reshape(-1, 3)
    parts.append(g)
    # 5. uniformly random bit patterns
    parts.append(rng.integers(0, 1 << 32, (1 << 19, 3), dtype=np.uint64).astype(np.uint32).view(np.float32))
    # 6. mantissas at the top of a binade (rsqrt14 rounds up to an exact power of two) for both exponent parities
    m = (np.uint32(0x7fffff) - np.arange(0, 4096, dtype=np.uint32))
    for e in (126, 127, 128, 129, 100, 101):
        v = ((np.uint32(e) << np.uint32(23)) | m).view(np.float32)
        parts.append(np.stack([v, np.zeros_like(v), np.zeros_like(v)], 1))       # L1 = a, L2 = 0
        parts.append(np.stack([v, v * np.float32(0.5), v], 1))
    return np.ascontiguousarray(np.concatenate(parts).astype(np.float32, copy=False))


@pytest.mark.parametrize("fold,bits", [("filters_2x/filters_highres", 8), ("filters_2x/filters_lowres", 10)])
def test_device_hash_matches_oracle_on_adversarial_triples(fold, bits):
    import oracle_py as O
    import raisr_hip as R
    abd = _triples(1234 + bits)
    dev = R.RaisrDevice(0, hooks=True)
    try:
        dev.set_model_from_folder(folder(fold), bits, 1)
        for flavour, avx2 in ((R.HASH_AVX512, 0), (R.HASH_AVX2, 1)):
            p = O.make_pass(O.Model(folder(fold), bits, 1), bits, False, 1 if avx2 else 2)
            want = O.hash_array(abd, p, avx2)
            got = dev.debug_hash(abd, 0, flavour)
            bad = np.nonzero(want != got)[0]
            assert bad.size == 0, (flavour, bad.size, abd[bad[:5]].view(np.uint32), want[bad[:5]], got[bad[:5]])
            assert want.max() <= 215 and len(np.unique(want)) > 150        # the sweep reaches most buckets
    finally:
        dev.close()
