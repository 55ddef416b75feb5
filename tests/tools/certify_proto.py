"""Prototype + validation of the approximate-then-certify hash (DESIGN.md s5 "certified hashing").

TEST INFRASTRUCTURE (uses the CPU oracle).  numpy model of what the HIP kernel does:
  1. approximate structure tensor (a', b', d') from a SEPARABLE 11+11-tap Gaussian on the per-pixel gradient products,
  2. real-arithmetic hash quantities at (a', b', d') with native sqrt/division,
  3. rigorous perturbation bounds -> bucket certified or not,
and the check that every certified bucket equals the exact oracle hash of the exact tensor.
Run:  python tests/tools/certify_proto.py [frame kinds...]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "video-super-resolution-library_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import oracle_py as O  # noqa: E402
import synth  # noqa: E402

U = 2.0 ** -24

GAUSS_Q = np.array([
    [7.76554e-05, 0.000239195, 0.0005738, 0.001072, 0.00155975, 0.00176743],
    [0.000239195, 0.000736774, 0.00176743, 0.00330199, 0.00480437, 0.00544406],
    [0.0005738, 0.00176743, 0.00423984, 0.00792107, 0.0115251, 0.0130596],
    [0.001072, 0.00330199, 0.00792107, 0.0147985, 0.0215317, 0.0243986],
    [0.00155975, 0.00480437, 0.0115251, 0.0215317, 0.0313284, 0.0354998],
    [0.00176743, 0.00544406, 0.0130596, 0.0243986, 0.0354998, 0.0402265]])


def literal_table():
    idx = [0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0]
    return GAUSS_Q[np.ix_(idx, idx)]


def rank1_fit():
    """u (11,) with u_i u_k ~ literal[i][k]; returns (u, eps_w = max relative deviation)."""
    lit = literal_table()
    # log-domain least squares start, then minimax polish by coordinate search
    lg = np.log(GAUSS_Q)
    # log q_ij = x_i + x_j
    A = []; y = []
    for i in range(6):
        for j in range(6):
            r = np.zeros(6); r[i] += 1; r[j] += 1
            A.append(r); y.append(lg[i, j])
    x = np.linalg.lstsq(np.array(A), np.array(y), rcond=None)[0]
    u6 = np.exp(x)

    def err(u6):
        return np.max(np.abs(np.outer(u6, u6) / GAUSS_Q - 1))
    best = err(u6)
    step = 1e-6
    for _ in range(200):
        improved = False
        for i in range(6):
            for s in (step, -step):
                t = u6.copy(); t[i] *= (1 + s)
                e = err(t)
                if e < best:
                    best, u6, improved = e, t, True
        if not improved:
            step /= 2
            if step < 1e-10:
                break
    u = u6[[0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0]]
    return u, float(np.max(np.abs(np.outer(u, u) / lit - 1)))


def exact_tensor(lr, bits):
    L = O.lib()
    L.ora_tensor_plane.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
    h, w = lr.shape
    lr16 = np.ascontiguousarray(lr, np.uint16)
    a = np.zeros((h, w), np.float32); b = np.zeros((h, w), np.float32); d = np.zeros((h, w), np.float32)
    L.ora_tensor_plane(lr16.ctypes.data, w, h, bits, a.ctypes.data, b.ctypes.data, d.ctypes.data)
    return a, b, d


def approx_tensor(lr, bits, u):
    """fp32 separable tensor, V pass then H pass (mul+add per tap: an upper bound on the fma rounding of the kernel)."""
    f = np.float32
    maxv = {8: 255.0, 10: 1023.0, 16: 65535.0}[bits]
    nf = f(1.0) / f(f(f(maxv) * f(maxv)) * f(2.0) * f(2.0))
    us = (np.sqrt(np.float64(nf)) * u).astype(np.float32)      # sqrt(NF) folded into both passes
    L = lr.astype(np.float32)
    h, w = L.shape
    gx = np.zeros_like(L); gy = np.zeros_like(L)
    gx[1:-1, :] = L[2:, :] - L[:-2, :]
    gy[:, 1:-1] = L[:, 2:] - L[:, :-2]
    out = []
    for P in (gx * gx, gx * gy, gy * gy):
        V = np.zeros_like(P)
        for i in range(11):                       # V(y, x) = sum_i us_i P(y + i - 5, x)
            sh = np.zeros_like(P)
            lo, hi = max(0, 5 - i), min(h, h + 5 - i)
            sh[lo:hi, :] = P[lo + i - 5:hi + i - 5, :]
            V = (V + us[i] * sh).astype(np.float32)
        Hh = np.zeros_like(P)
        for k in range(11):
            sh = np.zeros_like(P)
            lo, hi = max(0, 5 - k), min(w, w + 5 - k)
            sh[:, lo:hi] = V[:, lo + k - 5:hi + k - 5]
            Hh = (Hh + us[k] * sh).astype(np.float32)
        out.append(Hh)
    return out


def certify(a, b, d, P, eps, E14):
    """Returns (bucket, certified mask).  a, b, d: approximate tensor (float32 arrays); all math in float32 like the kernel,
    bounds per DESIGN.md s5.  P: OraPass with thresholds."""
    f = np.float32
    a = a.astype(f); b = b.astype(f); d = d.astype(f)
    qangle = f(P.qangle); qs = [f(P.qstr[0]), f(P.qstr[1])]; qc = [f(P.qcoh[0]), f(P.qcoh[1])]
    with np.errstate(all="ignore"):
        T = a + d
        m = f(0.5) * (a - d)
        R = m * m + b * b
        s = np.sqrt(R)
        hT = f(0.5) * T
        L1 = hT + s
        # L2 = hT - s = (ad - b^2) / L1 without cancellation
        det = a * d - b * b
        L2 = det / L1
        # bounds
        E_s = f(1.42 * eps) * T + f(2e-7 + eps * eps) * T * T / s + f(E14 * 1.05) * s
        ok = (T > 0) & (s > 0) & (E_s <= f(0.25) * s)
        E_L = E_s + f(0.5 * eps + 3 * U) * T                       # L1, L2, xx
        # det has its own cancellation when computed in fp32: |err| <= 3u(ad + b^2) -> L2 error 3u*T^2/2/L1 <= 3u T
        E_L2 = E_L + f(4 * U) * T
        # strength
        c_str = (np.abs(L1 - qs[0]) > E_L) & (np.abs(L1 - qs[1]) > E_L)
        si = (qs[0] <= L1).astype(np.int32) + (qs[1] <= L1)
        # coherence
        okc = L2 > f(2.0) * E_L2
        t = np.sqrt(L2 / L1)
        coh = (f(1) - t) / (f(1) + t)
        rho = f(0.55) * (E_L2 / (L2 - E_L2) + E_L / (L1 - E_L)) + f(2.4 * E14)
        r1t = f(1) / (f(1) + t)
        slope = np.where((rho <= f(0.0625)) & (t <= f(1.0)), f(2.07) * (r1t * r1t), f(2.0))   # round 5: 1 / (1 + t)^2 of d coh / dt kept (was bounded by 1)
        dcoh = (slope * t) * rho + f(2e-6)
        c_coh = okc & (np.abs(coh - qc[0]) > dcoh) & (np.abs(coh - qc[1]) > dcoh)
        ci = (qc[0] <= coh).astype(np.int32) + (qc[1] <= coh)
        # angle
        E_b = f(0.5 * eps) * T
        bsure = np.abs(b) > E_b
        ay = np.abs(b) + f(1e-10)
        xx = np.where(m >= 0, m + s, b * b / (s - m))               # (s - m)(s + m) = b^2
        E_xx = E_L
        E_ay = E_b + f(U) * ay
        D = xx + ay - E_xx - E_ay
        okx = (xx > f(2.0) * E_xx) & (D > 0)
        rr = (xx - ay) / (xx + ay)
        drr = f(2) * ((ay + E_ay) * E_xx + (xx + E_xx) * E_ay) / (D * D) + f(4 * U)
        ang_raw = (f(0.1963) * rr * rr + f(-0.9817)) * rr + f(np.pi / 4)
        dang = drr + f(1.5e-6)
        ang = np.where(b < 0, -ang_raw, ang_raw)
        ang = np.where(ang < 0, ang + f(3.141592653), ang)
        q = ang * qangle
        dq = qangle * dang + f(2e-5)
        k = np.clip(np.floor(q), 0, 23)
        fr = q - k
        c_ang = okx & bsure & (np.abs(ang_raw) > dang) & ((k < 1) | (fr > dq)) & ((k > 22) | (f(1) - fr > dq))
        ai = k.astype(np.int32)
        cert = ok & c_str & c_coh & c_ang
        certify.parts = dict(ok=ok, c_str=c_str, okc=okc, c_coh=c_coh, okx=okx, bsure=bsure, c_ang=c_ang,
                             angraw=(np.abs(ang_raw) > dang))
        zero = (T == 0)                                              # flat window: X == X' == 0 exactly -> bucket of (0,0,0)
        bucket = ai * 9 + si * 3 + ci
    return bucket, cert, zero


def sqrt14_max_rel_err():
    """max |sqrt14(x)/sqrt(x) - 1| over all inputs in [1, 4) at 2^-17 mantissa resolution (the instructions ignore
    lower bits)."""
    L = O.lib()
    n = 1 << 18
    xs = (np.float64(1.0) + np.arange(n) / n * 3.0).astype(np.float32)
    out = np.empty(n, np.float32)
    for i in range(n):
        out[i] = L.ora_x86_rcp14(L.ora_x86_rsqrt14(float(xs[i])))
    return float(np.max(np.abs(out.astype(np.float64) / np.sqrt(xs.astype(np.float64)) - 1)))


def kernel_weights():
    """What make_sep() in csrc/raisr_kernels.hip does: u_i = sqrt(literal_ii) (the NF factor cancels in the ratio), eps_w measured
    on the products, eps = 1.05 (eps_w + 48 u)."""
    lit = literal_table()
    u = np.sqrt(np.diag(lit))
    eps_w = float(np.max(np.abs(np.outer(u, u) / lit - 1)))
    return u, eps_w, 1.05 * (eps_w + 48 * U)


def run(kinds, w=480, h=270, bits=8, folder="filters_2x/filters_highres", eps=None, E14=None, legacy=False, quiet=False):
    """Returns {kind: (fallback fraction, false certifications)}.  Mirrors the kernel: sqrt-diagonal weights, E = 1.0e-4 for
    the AVX-512 flavour and 6.5e-4 for the AVX2 flavour (legacy=True)."""
    u, eps_w, eps_k = kernel_weights()
    if eps is None:
        eps = eps_k
    if not quiet:
        print(f"eps_w = {eps_w:.3e}  eps = {eps:.3e}")
    if E14 is None:
        E14 = 6.5e-4 if legacy else 1.0e-4
    results = {}
    m = O.Model(os.path.join(ROOT, folder), bits, 1)
    P = O.make_pass(m, bits, False, O.ASM_AVX512)
    for kind in kinds:
        if kind == "natural":
            y = synth.natural_y(w, h, bits, seed=12345)
        elif kind == "random":
            y = synth.random_y(w, h, bits, seed=777)
        elif kind == "smooth":
            yy, xx = np.mgrid[0:h, 0:w]
            y = np.clip(60 + 0.2 * xx + 0.1 * yy + 20 * np.sin(xx / 17.0) * np.cos(yy / 23.0), 0, 255).astype(np.uint8)
        elif kind == "edges":
            y = np.full((h, w), 40, np.uint8); y[:, w // 3:] = 200; y[h // 2:, :] //= 2; y[h // 4:h // 3, :] = 90
        else:
            y = synth.FRAME_KINDS[kind](w, h, bits)
        lr = O.resize(y, 2 * w, 2 * h)
        a, b, d = exact_tensor(lr, bits)
        a2, b2, d2 = approx_tensor(lr, bits, u)
        z = (slice(6, 2 * h - 6), slice(6, 2 * w - 6))
        ea = np.abs(a2[z] - a[z]) / np.maximum(a[z], 1e-30)
        ed = np.abs(d2[z] - d[z]) / np.maximum(d[z], 1e-30)
        eb = np.abs(b2[z] - b[z]) / np.maximum(0.5 * (a[z] + d[z]), 1e-30)
        nz = (a[z] > 0) & (d[z] > 0)
        dev = (float(ea[a[z] > 0].max()) if (a[z] > 0).any() else 0.0, float(ed[d[z] > 0].max()) if (d[z] > 0).any() else 0.0,
               float(eb[nz].max()) if nz.any() else 0.0)
        if not quiet:
            print(f"[{kind}] max rel dev a {dev[0]:.2e} d {dev[1]:.2e} b/(T/2) {dev[2]:.2e}")
        abd = np.stack([a[z].ravel(), b[z].ravel(), d[z].ravel()], 1)
        hx = O.hash_array(abd, P, legacy).reshape(a[z].shape)
        bucket, cert, zero = certify(a2[z], b2[z], d2[z], P, eps, E14)
        h0 = int(O.hash_array(np.zeros((1, 3), np.float32), P, legacy)[0])
        bucket = np.where(zero, h0, bucket); cert = cert | zero
        if os.environ.get("DIAG"):
            for k2, v in certify.parts.items():
                print(f"      fail {k2}: {100 - (v | zero).mean() * 100:.2f}%")
        wrong = cert & (bucket != hx)
        results[kind] = (1.0 - float(cert.mean()), int(wrong.sum()), max(dev))
        if not quiet:
            print(f"[{kind}] certified {cert.mean() * 100:.2f}%  fallback {100 - cert.mean() * 100:.2f}%  FALSE-CERTIFIED {int(wrong.sum())}"
                  f"   (approx bucket == exact on {np.mean(bucket == hx) * 100:.2f}% of all)")
        if wrong.any() and not quiet:
            idx = np.argwhere(wrong)[:5]
            for r, c in idx:
                print("   ", (r, c), a[z][r, c], b[z][r, c], d[z][r, c], "approx", a2[z][r, c], b2[z][r, c], d2[z][r, c], bucket[r, c], hx[r, c])
    return results


if __name__ == "__main__":
    kinds = sys.argv[1:] or ["natural", "random", "smooth", "edges", "constant", "checker"]
    run(kinds)

