"""CPU: the error bounds of the certified hash stage (docs/CERTIFY.md, csrc/kernels_hash_certify.h: approx_hash) MACHINE-CHECKED with
outward-rounded interval arithmetic.

The certified stage keeps an approximate bucket only if bounds E_s, E_L, E_L2, dcoh, dang, dq -- evaluated from the APPROXIMATE tensor
(a', b', d') -- prove that the reference's own instruction sequence (GetHashValue_AVX512_32f_16Elements, Raisr_AVX512.cpp:175-258)
lands in the same bucket for EVERY exact tensor (a, b, d) compatible with the approximation.  docs/CERTIFY.md derives those bounds by
hand; this file re-derives each inequality as interval arithmetic over a subdivision of the normalised domain (T' = 1, (m', b') in
the disc of positive semi-definite tensors, the additive constants 1e-10 / 1e-17 as scale parameters) and fails if any box cannot be
proved.  Every floating-point operation of either side is modelled as (exact result) * (1 + theta), |theta| <= u = 2^-24 (v_sqrt_f32 /
v_rcp_f32: 2u, the documented 1 ulp); the x86 approximation instructions as sqrt(x) (1 + e), |e| <= E (enumerated in
tests/test_certify_bounds.py).  The constants are READ FROM THE SOURCES (approx_hash, make_sep), not retyped.

Assumptions (stated, not proved here): no underflow in the modelled operations (tensors of real frames: T >= 4e-15, docs/CERTIFY.md
s1), the vendor's 1-ulp sqrt / rcp, the enumerated E.  Interval arithmetic is sound whatever the subdivision; the subdivision only
has to be fine enough for the proof to go through."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = 2.0 ** -24


# ---- outward-rounded interval arithmetic on numpy arrays (binary64; every operation widened by one ulp on each side) -------------
def _dn(x):
    return np.nextafter(x, -np.inf)


def _up(x):
    return np.nextafter(x, np.inf)


class IV:
    __slots__ = ("lo", "hi")

    def __init__(self, lo, hi=None):
        lo = np.asarray(lo, np.float64)
        hi = lo if hi is None else np.asarray(hi, np.float64)
        self.lo, self.hi = np.broadcast_arrays(lo, hi)
        assert np.all(self.lo <= self.hi)

    @staticmethod
    def of(x):
        return x if isinstance(x, IV) else IV(x)

    def __add__(self, o):
        o = IV.of(o)
        return IV(_dn(self.lo + o.lo), _up(self.hi + o.hi))
    __radd__ = __add__

    def __neg__(self):
        return IV(-self.hi, -self.lo)

    def __sub__(self, o):
        o = IV.of(o)
        return IV(_dn(self.lo - o.hi), _up(self.hi - o.lo))

    def __rsub__(self, o):
        return IV.of(o) - self

    def __mul__(self, o):
        o = IV.of(o)
        p = np.stack(np.broadcast_arrays(self.lo * o.lo, self.lo * o.hi, self.hi * o.lo, self.hi * o.hi))
        return IV(_dn(p.min(0)), _up(p.max(0)))
    __rmul__ = __mul__

    def inv(self):
        assert np.all((self.lo > 0) | (self.hi < 0)), "division by an interval containing zero"
        return IV(_dn(1.0 / self.hi), _up(1.0 / self.lo))

    def __truediv__(self, o):
        return self * IV.of(o).inv()

    def __rtruediv__(self, o):
        return IV.of(o) * self.inv()

    def sqrt(self):
        assert np.all(self.lo >= 0)
        return IV(np.maximum(_dn(np.sqrt(self.lo)), 0.0), _up(np.sqrt(self.hi)))

    def sq(self):
        a, b = self.lo * self.lo, self.hi * self.hi
        lo = np.where((self.lo <= 0) & (self.hi >= 0), 0.0, np.minimum(a, b))
        return IV(np.maximum(_dn(lo), 0.0), _up(np.maximum(a, b)))

    def abs_hi(self):
        return np.maximum(np.abs(self.lo), np.abs(self.hi))

    def cos(self):                                            # only used on boxes inside one monotone piece: enclosure from the end points + 1e-15
        c = np.stack([np.cos(self.lo), np.cos(self.hi)])
        return IV(c.min(0) - 1e-15, c.max(0) + 1e-15)

    def sin(self):
        c = np.stack([np.sin(self.lo), np.sin(self.hi)])
        return IV(c.min(0) - 1e-15, c.max(0) + 1e-15)


def sym(x):
    """[-x, x]"""
    x = np.asarray(x, np.float64)
    return IV(-x, x)


RND = sym(U)                                                  # theta of one fp32 rounding; every use is an independent occurrence
RND2 = sym(2 * U)                                             # v_sqrt_f32 / v_rcp_f32 (1 ulp)


def rnd(x):
    return x * (1.0 + RND)


# ---- the constants, from the sources ----------------------------------------------------------------------------------------------
def _constants():
    hip = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "device_abi.hip")).read()
    hdr = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "kernels_hash_certify.h")).read()
    ms = hip[hip.index("SepW make_sep("):]
    ms = ms[:ms.index("\n}\n")]
    ah = hdr[hdr.index("__device__ __forceinline__ bool approx_hash("):]
    ah = ah[:ah.index("\n}\n")]

    def one(pattern, text, what):
        m = re.search(pattern, text)
        assert m, f"cannot find {what} in the sources: the test must follow the code"
        return float(m.group(1))
    C = {}
    C["eps_mul"] = one(r"const double eps = ([0-9.]+) \* \(eps_w \+", ms, "eps factor")
    C["eps_u"] = one(r"\(eps_w \+ ([0-9.]+) \* u\)", ms, "eps roundings")
    C["es1"] = one(r"S\.es1 = \(float\)\(([0-9.e-]+) \* eps\)", ms, "es1")
    C["es2"] = one(r"S\.es2 = \(float\)\(([0-9.e-]+) \+ eps \* eps\)", ms, "es2")
    C["eEL_u"] = one(r"S\.eEL = \(float\)\(0\.5 \* eps \+ ([0-9.]+) \* u\)", ms, "eEL")
    assert re.search(r"S\.eEb = \(float\)\(0\.5 \* eps\)", ms)
    C["E"] = [float(x) for x in re.search(r"const double E\[2\] = \{([0-9.e-]+), ([0-9.e-]+)\}", ms).groups()]
    C["e105"] = one(r"S\.e105\[f\] = \(float\)\(([0-9.]+) \* E\[f\]\)", ms, "e105")
    C["e24"] = one(r"S\.e24\[f\] = \(float\)\(([0-9.]+) \* E\[f\]\)", ms, "e24")
    C["s_frac"] = one(r"\(E_s <= ([0-9.]+)f \* s\)", ah, "E_s <= s/4")
    C["EL2_u"] = one(r"E_L2 = __builtin_fmaf\(([0-9.]+)f \* U1, T, E_L\)", ah, "E_L2")
    C["L2_min"] = one(r"ok &= L2 > ([0-9.]+)f \* E_L2", ah, "L2 > 2 E_L2")
    C["c055"] = one(r"__builtin_fmaf\(([0-9.]+)f, __builtin_fmaf\(E_L2,", ah, "0.55")
    m = re.search(r"const float slope = \(\(rho <= ([0-9.]+)f\) & \(t <= 1\.0f\)\) \? ([0-9.]+)f \* \(r1t \* r1t\) : ([0-9.]+)f;", ah)
    assert m, "cannot find the coherence slope in the sources: the test must follow the code"
    C["rho_max"], C["coh_mul"], C["coh_mul_plain"] = (float(x) for x in m.groups())
    C["coh_add"] = one(r"const float dcoh = __builtin_fmaf\(slope \* t, rho, ([0-9.e-]+)f\);", ah, "coherence slack")
    assert "const float rho = __builtin_fmaf(" in ah and "S.e24[fl]);" in ah and "const float r1t = __builtin_amdgcn_rcpf(1.0f + t);" in ah
    C["xx_min"] = one(r"\(xx > ([0-9.]+)f \* E_L\)", ah, "xx > 2 E_L")
    C["drr_u"] = one(r"rD \* rD, ([0-9.]+)f \* U1\)", ah, "drr slack")
    C["dang_add"] = one(r"const float dang = drr \+ ([0-9.e-]+)f;", ah, "dang slack")
    C["dq_add"] = one(r"const float dq = __builtin_fmaf\(Q\.qangle, dang, ([0-9.e-]+)f\);", ah, "dq slack")
    C["p3"] = one(r"__builtin_fmaf\(__builtin_fmaf\(([0-9.]+)f \* rr, rr,", ah, "cubic coefficient")
    C["p1"] = one(r"rr, rr, (-[0-9.]+)f\), rr, ONEQTR_PI\)", ah, "linear coefficient")
    assert "E_s = __builtin_fmaf(S.es1, T, __builtin_fmaf((S.es2 * T) * T, rs, S.e105[fl] * s))" in ah      # the shape of E_s this file models
    assert "E_L = __builtin_fmaf(S.eEL, T, E_s)" in ah and "E_b = S.eEb * T" in ah and re.search(r"E_ay = __builtin_fmaf\([0-9.]+f \* U1, ay, E_b\)", ah)
    return C


ETA_U = 2.001             # the reference's radicand: |rad - R| <= ETA_U u T^2 (test_reference_radicand_error; the kernel budgets 2e-7 = 3.36 u)
EPS_W_MAX = 5e-6          # measured 2.3e-6 for every bit depth; tests/test_certify_bounds.py bounds it by 5e-6


def _eps(C, eps_w):
    return C["eps_mul"] * (eps_w + C["eps_u"] * U)


def _true_dev(eps_w):
    """s1: |a - a'| <= dev a' (and d; |b - b'| <= dev T'/2): 16 roundings of the reference against eps_w and 23 roundings of the kernel."""
    return (1.0 + 16.0001 * sym(U)) / ((1.0 - sym(eps_w)) * (1.0 - 23.0001 * sym(U))) - 1.0


def test_eps_covers_both_tensors_rounding_and_the_rank1_fit():
    C = _constants()
    ew = np.linspace(0.0, EPS_W_MAX, 2001)
    dev = _true_dev(ew).abs_hi()
    assert np.all(dev <= _eps(C, ew) * (1 - 1e-9)), "eps = 1.05 (eps_w + 48 u) does not cover the two tensors' deviation"
    assert np.all(dev <= 0.97 * _eps(C, ew))                  # ... with the margin the later lemmas may use


def test_reference_radicand_error():
    """s3: rad = fl(fl(fl(T T) / 4) - fl(fl(a d) - fl(b b))) = R + eta, |eta| <= ETA_U u T^2 (2 u up to second-order terms), for a, d >= 0,
    b^2 <= a d (1 + 1e-4), T = 1."""
    a = np.linspace(0.0, 1.0, 4001)
    a = IV(a[:-1], a[1:])
    d = 1.0 - a
    ad = a * d
    # b^2 anywhere in [0, ad (1 + 1e-4)]
    bb = IV(np.zeros_like(ad.lo), ad.hi * (1 + 1e-4))
    Tc = (a + d) * (1.0 + RND)                               # (a + d itself is 1: the interval of a + (1 - a) is wider, harmless)
    Q = 0.25
    F3 = (1.0 + RND) * (1.0 + RND) * (1.0 + RND) - 1.0       # T rounded, squared (twice the factor), product rounded; / 4 exact
    F3 = ((1.0 + RND).sq() * (1.0 + RND)) - 1.0
    qQ = Q * F3
    F2 = (1.0 + RND) * (1.0 + RND) - 1.0
    DtD = ad * F2 - bb * F2
    q_minus_Dt = (Q + qQ) - ((ad - bb) + DtD)
    eta = qQ - DtD + RND * q_minus_Dt
    assert np.all(eta.abs_hi() <= ETA_U * U), float(eta.abs_hi().max() / U)
    del Tc


def _polar_boxes(n_rho, n_phi, rho_lo):
    edges = np.geomspace(rho_lo, 0.5 * (1 + 2e-5), n_rho + 1)
    ph = np.linspace(0.0, 2 * np.pi, n_phi + 1)                # n_phi a multiple of 4: no box straddles an extremum of cos / sin
    r0, p0 = np.meshgrid(edges[:-1], ph[:-1], indexing="ij")
    r1, p1 = np.meshgrid(edges[1:], ph[1:], indexing="ij")
    return IV(r0.ravel(), r1.ravel()), IV(p0.ravel(), p1.ravel())


def test_root_eigenvalues_and_xx():
    """s3-4: |s - s'| <= E_s, |L1 - L1'| <= E_L, |xx - xx'| <= E_L, |L2 - L2'| <= E_L2 wherever T' > 0, s' > 0, E_s <= s'/4 may hold."""
    C = _constants()
    for flav, E in enumerate(C["E"]):
        for eps_w in (0.0, 2.3e-6, EPS_W_MAX):
            eps = _eps(C, eps_w)
            dev = float(_true_dev(eps_w).abs_hi())
            es1, es2 = C["es1"] * eps, C["es2"] + eps * eps
            eEL = 0.5 * eps + C["eEL_u"] * U
            # the precondition needs es2 / s' <= s'/4: below rho = 2 sqrt(es2) nothing is certified
            rho, phi = _polar_boxes(260, 96, 1.9 * np.sqrt(es2))
            mp, bp = rho * phi.cos(), rho * phi.sin()          # m', b'  (T' = 1)
            ap, dp = 0.5 + mp, 0.5 - mp
            A = sym(dev)
            # exact tensor: a = a'(1 + alpha), d = d'(1 + beta), b = b' + gamma / 2
            dm = (ap * A - dp * A) * 0.5                        # m - m'
            db = A * 0.5                                        # b - b'
            dT = ap * A + dp * A                                # (a + d) - 1
            # reference: rad = R + eta; s = sqrt(rad)(1 + e)
            RmR = 2.0 * mp * dm + dm.sq() + 2.0 * bp * db + db.sq()
            eta = sym(ETA_U * U) * (1.0 + dT).sq()
            rad_d = RmR + eta                                   # rad - rho^2
            rad = rho.sq() + rad_d
            ok_dom = rad.lo > 0
            rad = IV(np.where(ok_dom, rad.lo, 1e-300), rad.hi)
            srad = rad.sqrt()
            s_ref_minus_rho = rad_d / (srad + rho) + sym(E) * srad
            # kernel: m'_k = fl(0.5 fl(a' - d')), bb = fl(b'^2), R' = fl(fma(m', m', bb)), s' = v_sqrt(R')
            Rk = (mp.sq() * (1.0 + RND).sq() + bp.sq() * (1.0 + RND)) * (1.0 + RND)
            sk = Rk.sqrt() * (1.0 + RND2)
            W = sk - rho                                        # s'_k - rho (relative a few u: the interval of rho cancels only partly --
            W = IV(np.minimum(W.lo, 0), np.maximum(W.hi, 0))    #  re-centre below)
            Wrel = ((1.0 + RND).sq() * (1.0 + RND)).sqrt() * (1.0 + RND2) - 1.0      # s'_k = rho (1 + Wrel) in the worst case
            S = s_ref_minus_rho - rho * Wrel                    # s_ref - s'_k
            # kernel's bound, smallest value it can take on the box (three fp32 fmas, v_rcp: relative 8 u at most)
            Tk = 1.0 + RND
            sk_hi = rho.hi * (1.0 + float(Wrel.hi)) * 1.0
            sk_lo = rho.lo * (1.0 + float(Wrel.lo))
            Es_lo = (es1 * float(Tk.lo) + es2 * float(Tk.lo) ** 2 / sk_hi + C["e105"] * E * sk_lo) * (1 - 8 * U)
            Es_hi = (es1 * float(Tk.hi) + es2 * float(Tk.hi) ** 2 / sk_lo + C["e105"] * E * sk_hi) * (1 + 8 * U)
            may_hold = Es_lo <= C["s_frac"] * sk_hi             # precondition E_s <= s'/4 possible somewhere in the box
            assert np.all(ok_dom | ~may_hold), "a box on which the stage may certify has a possibly negative radicand"
            bad = may_hold & (S.abs_hi() > Es_lo)
            assert not bad.any(), ("E_s", flav, eps_w, int(bad.sum()), float((S.abs_hi() / Es_lo)[may_hold].max()))
            # L1: reference fl(fl(T/2) + s), kernel fl(0.5 T' + s')
            Tc = (1.0 + dT) * (1.0 + RND)
            s_ref_hi = rho.hi + s_ref_minus_rho.abs_hi()
            L1_mag = 0.5 * (1 + dev) * (1 + U) + s_ref_hi
            L1d = 0.5 * (Tc - Tk) + S + RND * L1_mag + RND * (0.5 * (1 + U) + sk_hi)
            EL_lo = (eEL * float(Tk.lo) + Es_lo) * (1 - 2 * U)
            # (E_L uses the kernel's own E_s; where S is far below E_s the slack carries the u-level terms: both are evaluated per box)
            bad = may_hold & (L1d.abs_hi() > EL_lo)
            assert not bad.any(), ("E_L", flav, eps_w, int(bad.sum()))
            # xx: reference fl(L1 - d); kernel m' >= 0: fl(m'_k + s'_k), m' < 0: fl(bb_k rcp(fl(s'_k - m'_k))); the real value of the
            # kernel's xx' is m' + rho in both branches.  Everything in delta form (differences of the two sides, never of two boxes):
            #   L1 - d = m + s + 0.5 (a + d) theta_T + theta_L1 (T/2 + s),  m = m' + dm,  s = rho + (s_ref - rho)
            xx_mag = mp.abs_hi() + rho.hi + s_ref_minus_rho.abs_hi() + 0.5 * dev + 4 * U
            xx_ref_minus_true = dm + 0.5 * (1.0 + dT) * RND + s_ref_minus_rho + RND * L1_mag + RND * xx_mag
            k_plus = RND * mp.abs_hi() + rho * Wrel + RND * (mp.abs_hi() + sk_hi)           # branch m' >= 0: error of xx'_k against m' + rho
            # branch m' < 0: bb (1 rounding), s'_k - m'_k (both terms positive: relative error max(|Wrel|, u), then 1 rounding), v_rcp, product
            rel_minus = (1.0 + RND) * (1.0 + RND2) * (1.0 + RND) / ((1.0 + sym(float(Wrel.abs_hi()) + U)) * (1.0 + RND)) - 1.0
            xx_true_mag = np.maximum(np.abs((mp + rho).lo), np.abs((mp + rho).hi))
            k_minus = rel_minus * xx_true_mag
            neg = mp.lo < 0                                     # boxes where the m' < 0 branch may run
            pos = mp.hi >= 0
            kerr_hi = np.maximum(np.where(pos, k_plus.abs_hi(), 0.0), np.where(neg, k_minus.abs_hi(), 0.0))
            xxd_hi = xx_ref_minus_true.abs_hi() + kerr_hi
            bad = may_hold & (xxd_hi > EL_lo)
            assert not bad.any(), ("xx", flav, eps_w, int(bad.sum()), float((xxd_hi / EL_lo)[may_hold].max()))
            # L2: reference fl(fl(T/2) - s); kernel det' rcp(L1'), det' = fl(fma(a', d', -bb)): real value 1/2 - rho = (a'd' - b'^2) / (1/2 + rho)
            L2_ref_minus_true = 0.5 * (Tc - 1.0) - s_ref_minus_rho + RND * (0.5 * (1 + dev) * (1 + U) + s_ref_hi)
            # det_true = a'd' - b'^2 = 1/4 - rho^2 = (1/2 - rho)(1/2 + rho): no cancellation in this form
            L2true = 0.5 - rho
            L2true = IV(np.maximum(L2true.lo, 0.0), np.maximum(L2true.hi, 0.0))       # (rho <= 1/2 up to the kernel's own rounding of a', d')
            L1true = 0.5 + rho
            det_true_mag = (L2true * L1true).abs_hi() + 1e-5
            det_err = RND * bp.sq() + RND * (det_true_mag + bp.sq().hi * U)            # bb rounded; one rounding of the fma
            # L1'_k = (0.5 T'_k + s'_k)(1 + theta) = L1true (1 + lam), |lam| small
            lam = (0.5 * RND + rho * Wrel) / L1true + RND * (1.0 + sym(4 * U))
            F = (1.0 + RND2) * (1.0 + RND)                      # v_rcp, product
            # L2'_k - L2true = (det_true + det_err) F / (L1true (1 + lam)) - det_true / L1true = L2true (F / (1 + lam) - 1) + det_err F / (L1true (1 + lam))
            L2k_minus_true = L2true * (F / (1.0 + lam) - 1.0) + det_err * F / (L1true * (1.0 + lam))
            L2d = L2_ref_minus_true - L2k_minus_true
            EL2_lo = (C["EL2_u"] * U * float(Tk.lo) + EL_lo) * (1 - 2 * U)
            bad = may_hold & (L2d.abs_hi() > EL2_lo)
            assert not bad.any(), ("E_L2", flav, eps_w, int(bad.sum()), float((L2d.abs_hi() / EL2_lo)[may_hold].max()))
            del W, Es_hi


def test_coherence_bound():
    """s5: given |L1 - L1'| <= E_L = x1 L1', |L2 - L2'| <= E_L2 = x2 L2' with x2 <= 1/2 (the stage requires L2' > 2 E_L2), the kernel's
    rho = 0.55 (x2 / (1 - x2) + x1 / (1 - x1)) + 2.4 E bounds |t_ref - t_k| / t' for every t' = sqrt(L2'/L1') (step 1); and with
    rho <= 1/16 and coh = (1 - t) / (1 + t):  |coh - coh_k| <= 2.07 t_k rho r1t^2 + 2e-6 as the kernel evaluates it (step 2);
    2 t_k rho + 2e-6 for larger rho."""
    C = _constants()
    x = np.concatenate([[0.0], np.geomspace(1e-9, 1.0, 361)])
    for flav, E in enumerate(C["E"]):
        x1lo, x2lo = np.meshgrid(x[:-1], x[:-1], indexing="ij")
        x1hi, x2hi = np.meshgrid(x[1:], x[1:], indexing="ij")
        keep = (x1lo.ravel() < 0.3) & (x2lo.ravel() < 1.0 / C["L2_min"])
        x1 = IV(x1lo.ravel()[keep], np.minimum(x1hi.ravel()[keep], 0.3))
        x2 = IV(x2lo.ravel()[keep], np.minimum(x2hi.ravel()[keep], 1.0 / C["L2_min"]))
        X1, X2 = IV(-x1.hi, x1.hi), IV(-x2.hi, x2.hi)
        # t_ref / t'_real = sqrt((1 + X2) / (1 + X1)) (1 + e2) / (1 + e1); the kernel's t_k = t' (1 + 3.6 theta)
        g = ((1.0 + X2) / (1.0 + X1)).sqrt() * (1.0 + sym(E)) / (1.0 + sym(E))
        tk_rel = ((1.0 + RND2) * (1.0 + RND)).sqrt() * (1.0 + RND2)
        dt_over_t = g - tk_rel                                 # (t_ref - t_k) / t'
        # smallest value of the kernel's rho (three fp32 operations + two v_rcp on top of exact inputs: relative 10 u), times
        # t_k / t' >= 1 - 4 u
        rho_lo = (C["c055"] * (x2.lo / (1.0 - x2.lo) + x1.lo / (1.0 - x1.lo)) + C["e24"] * E) * (1 - 14 * U)
        assert np.all(dt_over_t.abs_hi() <= rho_lo), ("coherence, step 1", flav, float((dt_over_t.abs_hi() / rho_lo).max()))
    # step 2.  d := |t_ref - t_k| <= t_k rho (step 1), rho <= rho_max, so 1 + t_ref >= 1 + t_k - t_k rho_max:
    #   |f(t_ref) - f(t_k)| = 2 d / ((1 + t_ref)(1 + t_k)) <= 2 t_k rho / ((1 + t_k - t_k rho_max)(1 + t_k))
    # against the kernel's coh_mul t_k rho r1t^2 with r1t = v_rcp(fl(1 + t_k)) (2 u + 1 ulp) and three more roundings; per unit of
    # t_k rho, over t_k in [0, 1.001] (t_k = v_sqrt(L2 rL1) with L2 <= L1 up to roundings)
    tk = IV(np.linspace(0.0, 1.001, 4097)[:-1], np.linspace(0.0, 1.001, 4097)[1:])
    true_hi = (2.0 / ((1.0 + tk - tk * C["rho_max"]) * (1.0 + tk))).hi
    r1 = 1.0 / (1.0 + tk) * (1.0 + RND) * (1.0 + RND2)
    kern_lo = (C["coh_mul"] * (r1 * r1) * (1.0 + RND) * (1.0 + RND) * (1.0 + RND) * (1.0 + RND)).lo
    assert np.all(true_hi <= kern_lo), ("coherence, step 2", float((true_hi / kern_lo).max()))
    # rho > rho_max: denominators >= 1, |f(t_ref) - f(t_k)| <= 2 d <= 2 t_k rho (1 - 8 u) against the kernel's two products
    assert 2.0 * (1 - 8 * U) <= C["coh_mul_plain"] * (1 - 3 * U)
    # the additive slack covers what does not scale with t': four roundings of the reference's quotient, the 1e-17 in its denominator
    # (relative <= 1e-17 / sL1 <= 2.5e-10 for T >= 4e-15), the kernel's 1 - t, 1 + t, v_rcp, product
    assert 4 * U + 2.5e-10 + 6 * U <= C["coh_add"] * (1 - 1e-3)


def test_angle_bound():
    """s6: with |xx - xx'| <= E_L and ||b| - |b'|| <= dev T'/2 (the stage requires xx' > 2 E_L, |b'| > E_b, D = xx' + ay' - E_L - E_ay > 0):
    |ang_raw - ang_raw'| <= dang = 2 ((ay' + E_ay) E_L + (xx' + E_L) E_ay) / D^2 + 4 u + 1.5e-6 and |q - q'| <= qangle dang + 2e-5."""
    C = _constants()
    hdr = open(os.path.join(ROOT, "video-super-resolution-library_amd", "csrc", "kernels_hash_certify.h")).read()
    m = re.search(r"const float E_ay = __builtin_fmaf\(([0-9.]+)f \* U1, ay, E_b\);", hdr)
    assert m, "E_ay = fma(k u, ay, E_b) not found"
    k_ay = float(m.group(1))
    # (i) E_ay covers the true |ay - ay'|: ay = fl(|b| + 1e-10) on both sides, ||b| - |b'|| <= dev T'/2.  Per unit of T', kappa = 1e-10 / T' in
    #     [0, 2.5e4] (T' >= 4e-15), |b'| / T' in [0, 0.5001]
    #     true |ay - ay'| <= dev/2 + u ay_ref + u ay'_k,  ay_ref <= ay'(1 + u) + dev/2;   kernel: E_ay >= ((eps/2)(1 - u) + k u ay'_k)(1 - u),
    #     ay'_k >= ay'(1 - u).  Both sides are linear in ay': compare the constant parts and the coefficients of ay' (delta form).
    ay_max = 0.5001 + 2.5e4
    for eps_w in (0.0, 2.3e-6, EPS_W_MAX):
        eps, dev = _eps(C, eps_w), float(_true_dev(eps_w).abs_hi())
        const_true = IV(0.5 * dev) + U * (0.5 * dev) * (1 + U)
        coef_true = U * (1.0 + IV(U)) * (1.0 + IV(U)) + U * (1.0 + IV(U))           # u ay_ref + u ay'_k, per unit of ay'
        const_kern = IV(0.5 * eps) * (1 - U) * (1 - U)
        coef_kern = IV(k_ay * U) * (1 - U) * (1 - U)
        # const_true + coef_true ay' <= const_kern + coef_kern ay'  for every ay' in [0, ay_max]
        gap0 = const_kern - const_true
        assert float(gap0.lo) > 0
        worst = gap0 + (coef_kern - coef_true) * IV(0.0, ay_max)
        assert float(worst.lo) >= 0, (eps_w, float(worst.lo))
    # (ii) r(x, y) = (x - y)/(x + y):  r(x, y) - r(x', y') = 2 (dx y' - x' dy) / ((x + y)(x' + y'))  (identity, checked on a grid of
    #      points below), so with |dx| <= eL, |dy| <= eA, x + y >= D:  |...| <= 2 (eL y' + x' eA) / (D (x' + y'))  =: main;
    #      the kernel's  drr_main = 2 ((y' + eA) eL + (x' + eL) eA) / D^2,  and  main / drr_main = D_n / (1 + 2 eL eA / (eL y' + x' eA))
    #      with D_n = D / (x' + y') in (0, 1]: at most 1.  Intervals over the normalised box x' + y' = 1:
    pe = np.concatenate([[0.0], np.geomspace(1e-7, 1.0, 81)])
    ee = np.concatenate([[0.0], np.geomspace(1e-9, 1.0, 61)])
    P0, L0, A0 = np.meshgrid(pe[:-1], ee[:-1], ee[:-1], indexing="ij")
    P1, L1_, A1 = np.meshgrid(pe[1:], ee[1:], ee[1:], indexing="ij")
    p, eL, eA = IV(P0.ravel(), P1.ravel()), IV(L0.ravel(), L1_.ravel()), IV(A0.ravel(), A1.ravel())
    keep = (eL.lo + eA.lo < 1.0)
    p, eL, eA = (IV(v.lo[keep], v.hi[keep]) for v in (p, eL, eA))
    xq = 1.0 - p
    Dn = 1.0 - eL - eA
    Dn = IV(np.clip(Dn.lo, 0.0, 1.0), np.clip(Dn.hi, 0.0, 1.0))             # the stage requires D > 0
    N = eL * p + xq * eA
    g_lo = np.where(N.hi > 0, 2.0 * eL.lo * eA.lo / np.maximum(N.hi, 1e-300), 0.0)
    ratio_hi = Dn.hi / (1.0 + g_lo)
    assert np.all(ratio_hi <= 1.0)
    rng = np.random.default_rng(3)
    for _ in range(2000):                                                    # the identity itself, in exact rational arithmetic
        from fractions import Fraction as Fr
        x, y, x2, y2 = (Fr(int(v), 1000) for v in rng.integers(1, 5000, 4))
        assert (x - y) / (x + y) - (x2 - y2) / (x2 + y2) == 2 * ((x - x2) * y2 - x2 * (y - y2)) / ((x + y) * (x2 + y2))
    # (iii) the cubic and the roundings.  P(r) = (p3 r^2 + p1) r + pi/4 on [-1, 1]: |P'| = |3 p3 r^2 + p1| <= |p1| < 1.
    r = IV(-1.0, 1.0)
    dP = 3.0 * C["p3"] * r.sq() + C["p1"]
    p1 = abs(C["p1"])
    assert float(dP.abs_hi()) <= p1 * (1 + 1e-12) and p1 < 1.0
    # |rr - rr'| <= main + 8 u  (|r| <= 1: reference quotient 3 roundings, kernel 2 + v_rcp (2 u) + 1);  the cubic: 3 roundings of values <= 2
    # on either side.  Kernel's dang >= (drr_main (1 - 12 u) + drr_u u + dang_add)(1 - 2 u)  (its own fp32 evaluation):
    #   p1 (main + 8 u) + 12 u  <=  p1 drr_main + (8 p1 + 12) u  <=  drr_main (1 - 14 u) + (drr_u + dang_add / u)(1 - 2 u) u
    assert p1 <= 1.0 - 14 * U - 1e-9
    assert (8 * p1 + 12) * U <= (C["drr_u"] * U + C["dang_add"]) * (1 - 2 * U)
    # (iv) q = fl(ang qangle), ang = +-ang_raw (+ PI, one rounding, when negative): |q - q'| <= qangle dang + roundings of both sides
    qangle = 24.0 / 3.141592653 * (1 + 1e-6)
    assert 2 * (U * np.pi * qangle + U * 24.0) <= C["dq_add"] * (1 - 1e-3)
