"""-m gpu: the certified hash stage (k_hashfilter_ac).  The kernel computes the structure tensor approximately and keeps a
bucket only when rigorous bounds certify it; everything else takes the reference's exact instruction sequence.  Here:
self-check mode (every pixel ALSO takes the exact path; a certified bucket that differs is counted -- must be 0), output
bit-exact against the oracle, and the fraction of pixels that needed the exact path, per frame kind and flavour."""
import json
import os

import numpy as np
import pytest

from common import folder, oracle_y, dtype_for

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames(w, h, bits):
    import synth
    maxv = (1 << bits) - 1
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.clip(maxv * (0.25 + 0.0008 * xx + 0.0004 * yy + 0.08 * np.sin(xx / 17.0) * np.cos(yy / 23.0)), 0, maxv)
    edges = np.full((h, w), maxv * 40 // 255); edges[:, w // 3:] = maxv * 200 // 255; edges[h // 2:, :] //= 2
    edges[h // 4:h // 3, :] = maxv * 90 // 255
    dt = dtype_for(bits)
    return {"natural": synth.natural_y(w, h, bits, seed=12345), "random": synth.random_y(w, h, bits, seed=777),
            "checker": synth.checker_y(w, h, bits), "constant": synth.constant_y(w, h, bits),
            "smooth": smooth.astype(dt), "edges": edges.astype(dt)}


CASES = [  # (id, folder, ratio, bits, passes, mode, asm, full)
    ("2x_8b_avx512", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False),
    ("2x_8b_avx2", "filters_2x/filters_lowres", (2, 1), 8, 1, 1, 1, False),
    ("2x_10b_2p", "filters_2x/filters_highres", (2, 1), 10, 2, 1, 2, True),
    ("1.5x_8b_2p_m2", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 2, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_certified_buckets_equal_exact_buckets(case):
    import raisr_hip as R
    cid, fold, (rn, rd), bits, passes, mode, asm, full = case
    w, h = 372, 214                      # not a multiple of any tile size; several tiles in both directions
    ow, oh = w * rn // rd, h * rn // rd
    report = {}
    for kind, y in _frames(w, h, bits).items():
        ref = oracle_y(y, case)
        for check in (True, False):      # self-check mode (all pixels through both paths), then production mode
            dev = R.RaisrDevice(0, hooks=True)
            try:
                dev.set_model_from_folder(folder(fold), bits, passes)
                dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
                dev.certify_debug(True, check)
                out = np.zeros((oh, ow), dtype_for(bits))
                dev.process_host(np.ascontiguousarray(y), out)
                st = dev.certify_stats()
            finally:
                dev.close()
            assert st["pixels"] > 0, (cid, kind)
            assert st["mismatches"] == 0, (cid, kind, st)
            bad = np.argwhere(out != ref)
            assert bad.size == 0, (cid, kind, check, len(bad), bad[:5].tolist())
        report[kind] = round(st["uncertain"] / st["pixels"], 5)
    print("fallback fraction", cid, json.dumps(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"certify_fallback_{cid}.json"), "w") as f:
        json.dump(report, f)
    assert report["natural"] < 0.05 and (passes == 2 or report["constant"] == 0.0)   # pass 2 sees the 1-D rim pass 1 leaves at the frame margins


def test_exact_kernel_still_selectable(monkeypatch):
    """RAISR_HIP_CERTIFY=0 keeps the all-exact kernel (A/B switch): same bits."""
    import raisr_hip as R
    import synth
    monkeypatch.setenv("RAISR_HIP_CERTIFY", "0")
    case = CASES[0]
    y = synth.natural_y(200, 120, 8, seed=5)
    dev = R.RaisrDevice(0, hooks=True)
    try:
        dev.set_model_from_folder(folder(case[1]), 8, 1)
        dev.configure(200, 120, 400, 240, bits=8, passes=1, hash_variant=2)
        out = np.zeros((240, 400), np.uint8)
        dev.process_host(y, out)
        with pytest.raises(RuntimeError):
            dev.certify_stats()
    finally:
        dev.close()
    assert np.array_equal(out, oracle_y(y, case))


def test_full_size_self_check_c2():
    """BASELINE config 2 at full size, four frame kinds: self-check mode (every pixel through both paths) finds no certified
    bucket that differs from the exact one, and the production mode's output equals the oracle's."""
    import raisr_hip as R
    case = ("x", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False)
    w, h = 1920, 1080
    fr = _frames(w, h, 8)
    for kind in ("natural", "random", "smooth", "edges"):
        y = fr[kind]
        ref = oracle_y(y, case)
        for check in (True, False):
            dev = R.RaisrDevice(0, hooks=True)
            try:
                dev.set_model_from_folder(folder(case[1]), 8, 1)
                dev.configure(w, h, 2 * w, 2 * h, bits=8, passes=1, hash_variant=2)
                dev.certify_debug(True, check)
                out = np.zeros((2 * h, 2 * w), np.uint8)
                dev.process_host(np.ascontiguousarray(y), out)
                st = dev.certify_stats()
            finally:
                dev.close()
            assert st["mismatches"] == 0 and st["pixels"] == 3824 * 2148, (kind, check, st)
            assert np.array_equal(out, ref), (kind, check)
        print(kind, "fallback fraction", round(st["uncertain"] / st["pixels"], 5))


@pytest.mark.parametrize("flavour", [2, 1], ids=["avx512", "avx2"])
def test_certified_buckets_hold_on_the_whole_error_box(flavour):
    """The certification claim itself, away from images: for adversarial approximate tensors (a', b', d') -- all scales,
    near-isotropic, near-axis, near every threshold -- a bucket the device certifies must equal the oracle's exact hash of
    EVERY tensor in the error box |a-a'| <= eps a', |d-d'| <= eps d', |b-b'| <= eps (a'+d')/2.  The box is sampled at its 8
    corners, its centre and 7 random interior points per triple."""
    import oracle_py as O
    import raisr_hip as R
    rng = np.random.default_rng(20260928 + flavour)
    fold = "filters_2x/filters_highres"
    m = O.Model(folder(fold), 8, 1)
    P = O.make_pass(m, 8, False, O.ASM_AVX512)
    n = 400000
    scale = np.exp(rng.uniform(np.log(3e-10), np.log(0.3), n))
    # tensors (a, b, d) = rotation of eigenvalues (l1 >= l2 >= 0): anisotropy from degenerate to isotropic, all angles,
    # a third of the angles snapped next to a bucket boundary, a third of the strengths / coherences next to a threshold
    theta = rng.uniform(0, np.pi, n)
    snap = rng.random(n) < 0.33
    theta = np.where(snap, np.round(theta * 24 / np.pi) * np.pi / 24 + rng.normal(0, 3e-4, n), theta)
    ratio = np.where(rng.random(n) < 0.3, np.exp(rng.uniform(np.log(1e-7), 0, n)), rng.uniform(0, 1, n))     # l2 / l1
    qc = np.array([P.qcoh[0], P.qcoh[1]], np.float64)
    near_c = rng.random(n) < 0.25
    tq = (1 - qc[rng.integers(0, 2, n)]) / (1 + qc[rng.integers(0, 2, n)])
    ratio = np.where(near_c, np.clip(tq * tq * (1 + rng.normal(0, 5e-4, n)), 0, 1), ratio)
    l1 = scale
    qs = np.array([P.qstr[0], P.qstr[1]], np.float64)
    near_s = rng.random(n) < 0.25
    l1 = np.where(near_s, qs[rng.integers(0, 2, n)] * (1 + rng.normal(0, 5e-4, n)), l1)
    l2 = l1 * ratio
    cth, sth = np.cos(theta), np.sin(theta)
    a = l1 * cth * cth + l2 * sth * sth
    d = l1 * sth * sth + l2 * cth * cth
    b = (l1 - l2) * sth * cth
    abd = np.stack([a, b, d], 1).astype(np.float32)
    dev = R.RaisrDevice(0, hooks=True)
    try:
        dev.set_model_from_folder(folder(fold), 8, 1)
        dev.configure(64, 64, 128, 128, bits=8)
        bucket, cert, eps = dev.debug_approx_hash(abd, 0, R.HASH_AVX512 if flavour == 2 else R.HASH_AVX2)
    finally:
        dev.close()
    assert 1e-6 < eps < 2e-5, eps
    frac = cert.mean()
    assert 0.2 < frac < 0.98, frac                     # the adversarial mix must exercise both outcomes
    sel = np.nonzero(cert)[0]
    A = abd[sel].astype(np.float64)
    T = A[:, 0] + A[:, 2]
    bad_total = 0
    signs = [(sa, sb, sd) for sa in (-1, 1) for sb in (-1, 1) for sd in (-1, 1)] + [(0, 0, 0)]
    pts = [np.array(s3, np.float64) for s3 in signs] + [rng.uniform(-1, 1, 3) for _ in range(7)]
    for sg in pts:
        X = np.stack([A[:, 0] * (1 + sg[0] * eps * 0.999), A[:, 1] + sg[1] * eps * 0.999 * T / 2, A[:, 2] * (1 + sg[2] * eps * 0.999)], 1).astype(np.float32)
        hx = O.hash_array(X, P, flavour == 1)
        bad = hx != bucket[sel]
        bad_total += int(bad.sum())
        assert not bad.any(), (sg.tolist(), int(bad.sum()), abd[sel][bad][:3].tolist(), X[bad][:3].tolist(), bucket[sel][bad][:3].tolist(), hx[bad][:3].tolist())
    assert bad_total == 0


def test_zero_tensor_bucket_is_stable_across_context_lifetimes():
    """The bucket of the all-zero tensor is computed on the device when a model is set (k_debug_hash on the context stream).  Its
    input used to be cleared on the null stream, which nothing orders with a non-blocking stream: when the kernel won the race
    the bucket came out of stale memory and every flat-window pixel took a wrong filter (seen once as 16 k mismatches on a 1-px
    checkerboard, whose central differences all vanish).  Many create / set model / first frame cycles, flat-gradient content."""
    import raisr_hip as R
    import synth
    case = ("2x_10b_2p_m2", "filters_2x/filters_denoise", (2, 1), 10, 2, 2, 2, False)
    frames = [synth.checker_y(96, 64, 10), synth.constant_y(96, 64, 10)]
    refs = [oracle_y(y, case) for y in frames]
    junk = []
    for it in range(24):
        import torch
        junk.append(torch.full((1 + it,), float("nan"), device="cuda"))      # dirty a few small device allocations in between
        y, ref = frames[it % 2], refs[it % 2]
        dev = R.RaisrDevice(0, hooks=True)
        try:
            dev.set_model_from_folder(folder(case[1]), 10, 2)
            dev.configure(96, 64, 192, 128, bits=10, passes=2, mode=2, hash_variant=2)
            out = np.zeros((128, 192), np.uint16)
            dev.process_host(y, out)
        finally:
            dev.close()
        assert np.array_equal(out, ref), it
