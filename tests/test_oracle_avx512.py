"""CPU (hosts that execute AVX-512; skipped elsewhere): oracle/raisr_oracle_avx512.c -- the hand-vectorised twin of the fp32 pass
that bench.py times as the CPU baseline (own intrinsics, lane = pixel, the x86 approximation instructions as vectorised integer
models) -- against the scalar restatement raisr_oracle.c, bit for bit, and against the committed digests."""
import hashlib
import json
import os

import numpy as np
import pytest

from common import CASES, folder, oracle_y

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _intr(y, case, tie=0):
    import oracle_py as O
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    p1 = O.make_pass(O.Model(folder(fold), bits, 1), bits, full, asm)
    p2 = O.make_pass(O.Model(folder(fold), bits, 2), bits, full, asm) if passes == 2 else None
    return O.process_y_intrinsics(y, ow, oh, p1, p2, passes, mode, tie).astype(np.uint8 if bits == 8 else np.uint16)


def _need():
    import oracle_py as O
    if O.lib512() is None:
        pytest.skip("host does not execute AVX-512")


@pytest.mark.parametrize("case", [c for c in CASES if c[6] != 5], ids=lambda c: c[0])
def test_intrinsics_pass_equals_the_scalar_oracle(case):
    _need()
    import synth
    bits = case[3]
    maxv = (1 << bits) - 1
    for (w, h) in ((96, 64), (134, 50), (41, 37), (13, 20)):
        frames = {"natural": synth.natural_y(w, h, bits, seed=7), "random": synth.random_y(w, h, bits, seed=8),
                  "checker": synth.checker_y(w, h, bits), "constant": synth.constant_y(w, h, bits),
                  "extremes": (np.indices((h, w)).sum(0) % 2 * maxv).astype(np.uint8 if bits == 8 else np.uint16)}
        for kind, y in frames.items():
            ref, got = oracle_y(y, case), _intr(y, case)
            bad = np.argwhere(ref != got)
            assert bad.size == 0, (case[0], kind, (w, h), len(bad), bad[:4].tolist())


def test_intrinsics_pass_reproduces_the_committed_digests():
    _need()
    import synth
    want = json.load(open(os.path.join(ROOT, "tests/golden/oracle_digests.json")))
    n = 0
    for case in CASES:
        cid, bits, asm = case[0], case[3], case[6]
        if asm == 5:
            continue
        for nm, fr in (("natural", synth.natural_y(96, 64, bits, seed=4242)), ("random", synth.random_y(96, 64, bits, seed=99))):
            assert hashlib.sha256(_intr(fr, case).tobytes()).hexdigest() == want[f"{cid}/{nm}"], (cid, nm)
            n += 1
    assert n >= 20


def test_vector_models_of_the_approximation_instructions_on_special_values():
    """sqrt14_ps / sqrt_legacy_ps through the pass: a 16-bit full-range frame with flat, saturated and alternating regions drives
    zero tensors, negative radicands (NaN flow) and huge eigenvalues through both hash flavours."""
    _need()
    rng = np.random.default_rng(5)
    w, h = 120, 70
    y = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    y[:20, :] = 0; y[20:35, :] = 65535; y[35:50, ::2] = 0; y[35:50, 1::2] = 65535; y[50:, :40] = 12345
    import shutil, tempfile
    tmp = tempfile.mkdtemp()
    try:
        dst = os.path.join(tmp, "f16")
        shutil.copytree(folder("filters_2x/filters_highres"), dst)
        for stem in ("filterbin_2", "Qfactor_strbin_2", "Qfactor_cohbin_2"):
            for sfx in ("", "_2"):
                shutil.copyfile(os.path.join(dst, f"{stem}_10{sfx}"), os.path.join(dst, f"{stem}_16{sfx}"))
        for asm in (1, 2):
            case = ("x", dst, (2, 1), 16, 2, 1, asm, True)
            assert np.array_equal(oracle_y(y, case), _intr(y, case)), asm
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
