"""CPU: the constants the certified hash stage (csrc/raisr_kernels.hip: approx_hash / make_sep) relies on, and its numpy
model against the oracle.
  * sup |VRCP14(VRSQRT14(x)) / sqrt(x) - 1| <= 1.0e-4 and sup |RCPPS(RSQRTPS(x)) / sqrt(x) - 1| <= 6.5e-4, enumerated over
    every input step of the bit-exact instruction models (the results depend on the top mantissa bits and the exponent
    parity only, so two binades of steps are all there is);
  * the separable weights reproduce the literal table to eps_w < 5e-6;
  * tests/tools/certify_proto.py (the kernel's certification, in numpy) never certifies a bucket that differs from the
    oracle's exact hash of the exact tensor -- natural, random and degenerate frames, both hash flavours -- and the
    measured deviation of the separable tensor stays below the eps the bounds assume."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def _sup_rel_error(fn, mant_bits):
    """fn: scalar sqrt-approximation composition.  Steps of 2^-mant_bits in the mantissa, two binades [1,2) and [2,4)."""
    import oracle_py as O
    O.lib()
    n = 1 << mant_bits
    worst = 0.0
    for p in (1.0, 2.0):
        lo = (p * (1.0 + np.arange(n) / n)).astype(np.float32)
        hi = np.nextafter((p * (1.0 + (np.arange(n) + 1) / n)).astype(np.float32), np.float32(0))     # last float of the step
        lo1 = np.nextafter(lo, np.float32(8))            # exact powers of two are special-cased by the instructions: step interior starts one ulp later
        v_lo = np.array([fn(float(x)) for x in lo], np.float64)
        v_lo1 = np.array([fn(float(x)) for x in lo1], np.float64)
        v_hi = np.array([fn(float(x)) for x in hi], np.float64)
        assert np.array_equal(v_lo1, v_hi), "the composition is not constant within a step: enumerate finer"
        e = np.maximum(np.abs(v_lo1 / np.sqrt(lo1.astype(np.float64)) - 1), np.abs(v_lo1 / np.sqrt(hi.astype(np.float64)) - 1))
        e = np.maximum(e, np.abs(v_lo / np.sqrt(lo.astype(np.float64)) - 1))
        worst = max(worst, float(e.max()))
    return worst


def test_sqrt14_relative_error_bound():
    import oracle_py as O
    L = O.lib()
    sup = _sup_rel_error(lambda x: L.ora_x86_rcp14(L.ora_x86_rsqrt14(x)), 15)
    assert sup <= 1.0e-4, sup
    assert sup > 5e-5                                    # the constant is not grossly pessimistic either


def test_legacy_sqrt_relative_error_bound():
    import oracle_py as O
    L = O.lib()
    sup = _sup_rel_error(lambda x: L.ora_x86_rcp(L.ora_x86_rsqrt(x)), 12)
    assert sup <= 6.5e-4, sup


def test_separable_weights_reproduce_the_literal_table():
    import certify_proto as C
    u, eps_w, eps = C.kernel_weights()
    assert eps_w < 5e-6 and eps < 1e-5
    # in fp32, as the kernel holds them (sqrt(NF) folded into u), for every bit depth
    lit = C.literal_table()
    for maxv in (255.0, 1023.0, 65535.0):
        nf = np.float32(1.0) / np.float32(np.float32(np.float32(maxv) * np.float32(maxv)) * np.float32(4.0))
        w = (np.float64(nf) * lit).astype(np.float32)
        us = np.sqrt(np.diag(w).astype(np.float64)).astype(np.float32)
        r = np.outer(us.astype(np.float64), us.astype(np.float64)) / w.astype(np.float64) - 1
        assert np.max(np.abs(r)) < 5e-6, maxv


def test_certification_model_never_certifies_a_wrong_bucket():
    import certify_proto as C
    _, _, eps = C.kernel_weights()
    for legacy in (False, True):
        res = C.run(["natural", "random", "smooth", "edges", "constant", "checker"], w=240, h=136, legacy=legacy, quiet=True)
        for kind, (fallback, wrong, dev) in res.items():
            assert wrong == 0, (legacy, kind, wrong)
            assert dev < eps, (legacy, kind, dev)        # measured tensor deviation below the assumed eps
        assert res["natural"][0] < 0.03 and res["constant"][0] == 0.0
    res10 = C.run(["natural", "random"], w=200, h=120, bits=10, quiet=True)
    assert all(v[1] == 0 for v in res10.values())
