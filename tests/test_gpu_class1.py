"""-m gpu: the class-1 rule of the certified hash stage on the device (exactly one-dimensional windows; csrc/kernels_hash_certify.h,
docs/CERTIFY.md s9).

  * the sign table the context builds with the exact device models == the table derived from the ORACLE's hash of all 2^23 mantissas;
  * approx_hash's decision for class-1 tensors (a', 0, 0) / (0, 0, d'): every certified bucket equals the oracle's hash of (a, 0, 0)
    for a at both ends, the middle and random points of the box |a - a'| <= eps a', both flavours;
  * frames made of exactly one-dimensional windows (stripes, bars, block edges over flat ground, letterboxes; 8 / 10 / 16 bit, both
    hash flavours, 1.5x, two passes): HIP == oracle bit for bit, the self-check finds no certified-but-wrong bucket, and the rule does
    what it is for -- far fewer uncertified pixels and overflowed tiles than with RAISR_HIP_C1=0."""
import os

import numpy as np
import pytest

from common import CASES, folder, dtype_for, oracle_y
from test_class1 import reference_table

pytestmark = pytest.mark.gpu


def test_device_table_equals_the_oracle_derived_table():
    import raisr_hip as R
    tab, _ = reference_table()
    dev = R.RaisrDevice(0, hooks=True)
    got = dev.read_c1tab()
    dev.close()
    assert np.array_equal(got, tab), int((got != tab).sum())
    assert set(np.unique(tab)) == {1, 2, 3}


@pytest.mark.parametrize("flavour", [2, 1], ids=["avx512", "avx2"])
def test_class1_decisions_hold_on_the_whole_box(flavour):
    import oracle_py as O
    import raisr_hip as R
    fold = folder("filters_2x/filters_highres")
    dev = R.RaisrDevice(0, hooks=True)
    dev.set_model_from_folder(fold, 8, 1)
    dev.configure(192, 108, 384, 216, bits=8, passes=1, hash_variant=R.HASH_AVX512)
    P = O.make_pass(O.Model(fold, 8, 1), 8, False, O.ASM_AVX512)
    rng = np.random.default_rng(17)
    n = 200000
    a1 = np.exp(rng.uniform(np.log(5e-15), np.log(2.0), n)).astype(np.float32)
    a1[:2000] = np.concatenate([P.qstr[0] * (1 + rng.uniform(-3e-4, 3e-4, 1000)), P.qstr[1] * (1 + rng.uniform(-3e-4, 3e-4, 1000))]).astype(np.float32)  # L1 ~ a: near the strength thresholds
    abd = np.zeros((n, 3), np.float32)
    which = rng.integers(0, 2, n)
    abd[np.arange(n), 2 * which] = a1
    bucket, cert, eps = dev.debug_approx_hash(abd, 0, flavour)
    dev.close()
    assert 0.55 < cert.mean() <= 1.0 if flavour == 2 else cert.mean() > 0.97, cert.mean()
    legacy = flavour == 1
    a64 = a1.astype(np.float64)
    for frac in (-1.0, 1.0, 0.0, *rng.uniform(-1, 1, 5)):
        ex = np.zeros((n, 3), np.float32)
        av = (a64 * (1 + frac * eps)).astype(np.float32)
        av = np.where(np.abs(av.astype(np.float64) - a64) <= eps * a64, av, a1)       # the rounding of the end points may leave the box
        ex[np.arange(n), 2 * which] = av
        h = O.hash_array(ex, P, legacy)
        bad = cert & (h != bucket)
        assert not bad.any(), (frac, int(bad.sum()), abd[bad][:3], bucket[bad][:3], h[bad][:3])


def _one_dimensional_frames(w, h, bits):
    """frames whose windows are (mostly) exactly one-dimensional: every gy = 0 or every gx = 0"""
    maxv = (1 << bits) - 1
    dt = np.uint8 if bits == 8 else np.uint16
    rng = np.random.default_rng(1000 + bits)
    lo, hi = 16 * maxv // 255, 235 * maxv // 255
    yy, xx = np.mgrid[0:h, 0:w]
    out = {}
    col = rng.integers(lo, hi + 1, w)
    out["vertical stripes (random column levels)"] = np.broadcast_to(col, (h, w)).astype(dt)
    row = rng.integers(lo, hi + 1, h)
    out["horizontal bars (random row levels)"] = np.broadcast_to(row[:, None], (h, w)).astype(dt)
    blk = rng.integers(lo, hi + 1, ((h + 7) // 8, (w + 7) // 8))
    out["flat 8x8 blocks (a decoded low-bitrate picture)"] = np.kron(blk, np.ones((8, 8), int))[:h, :w].astype(dt)
    lb = np.full((h, w), lo, dt); lb[h // 5:h - h // 5, :] = (lo + (hi - lo) * (0.5 + 0.4 * np.sin(xx[h // 5:h - h // 5] / 9.0) * np.cos(yy[h // 5:h - h // 5] / 7.0))).astype(dt)
    out["letterbox around a smooth picture"] = lb
    ramp = (lo + (xx // 3) % (hi - lo)).astype(dt)
    out["staircase ramp in x"] = ramp
    step = np.full((h, w), lo + 3, dt); step[:, w // 2:] = lo + 4
    out["one-LSB vertical step"] = step
    return out


C1_CASES = [c for c in CASES if c[6] != 5 and c[0] in ("2x_highres_8b_1p_avx512", "2x_lowres_8b_1p_avx2", "2x_highres_8b_2p_m1", "2x_highres_10b_1p",
                                                     "1.5x_denoise_8b_2p_m2", "2x_highres_10b_2p_m1_full", "2x_denoise_10b_2p_m2")]


def _run(y, case, check, c1=True):
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    old = os.environ.get("RAISR_HIP_C1")
    if not c1:
        os.environ["RAISR_HIP_C1"] = "0"
    try:
        dev = R.RaisrDevice(0, hooks=True)
    finally:
        if not c1:
            if old is None:
                del os.environ["RAISR_HIP_C1"]
            else:
                os.environ["RAISR_HIP_C1"] = old
    dev.set_model_from_folder(folder(fold), bits, passes)
    dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
    dev.certify_debug(True, check)
    out = np.zeros((oh, ow), dtype_for(bits))
    dev.process_host(np.ascontiguousarray(y), out)
    st = dev.certify_stats()
    dev.close()
    return out, st


@pytest.mark.parametrize("case", C1_CASES, ids=[c[0] for c in C1_CASES])
def test_one_dimensional_frames_bit_exact_and_self_checked(case):
    bits = case[3]
    for name, y in _one_dimensional_frames(416, 240, bits).items():
        ref = oracle_y(y, case)
        got, st = _run(y, case, check=True)
        bad = np.argwhere(ref != got)
        assert bad.size == 0, f"{case[0]} / {name}: {len(bad)} mismatching pixels, first {bad[:5].tolist()}"
        assert st["mismatches"] == 0, (case[0], name, st)


def test_class1_does_what_it_is_for():
    """flat 8 x 8 blocks at 1080p -> 4K (43 % of the pixels see exactly one block edge: class 1; the corners of the blocks stay generic):
    with the rule the uncertified share drops by at least a quarter in the AVX-512 flavour (72 % of the class is certifiable) and by at
    least 40 % in the AVX2 flavour (all of it is); a JPEG-like picture -- the same blocks under a smooth picture, where most tiles are NOT
    hopeless -- loses most of its overflowed tiles.  Same bits either way."""
    blocks = _one_dimensional_frames(1920, 1080, 8)["flat 8x8 blocks (a decoded low-bitrate picture)"]
    yy, xx = np.mgrid[0:1080, 0:1920]
    smooth = (126 + 90 * np.sin(xx / 37.0) * np.cos(yy / 29.0))
    mixed = np.where(((xx // 160) + (yy // 120)) % 2 == 0, blocks, smooth).astype(np.uint8)       # half the picture in flat blocks
    for cid, drop in (("2x_highres_8b_1p_avx512", 0.75), ("2x_lowres_8b_1p_avx2", 0.60)):
        case = next(c for c in CASES if c[0] == cid)
        for name, y in (("blocks", blocks), ("mixed", mixed)):
            out_on, on = _run(y, case, check=False)
            out_off, off = _run(y, case, check=False, c1=False)
            assert np.array_equal(out_on, out_off), (cid, name)
            assert on["tiles"] == off["tiles"] == 8100                       # 3840 x 2160 in 64 x 16 tiles
            assert on["uncertain"] < drop * off["uncertain"], (cid, name, on, off)
            assert on["tiles_overflow"] <= off["tiles_overflow"], (cid, name, on, off)
