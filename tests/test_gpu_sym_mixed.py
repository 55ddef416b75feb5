"""-m gpu: the symmetric filter stage on banks with non-palindromic rows (csrc/kernels_filter.h, filter_phase<.., SYM>).

The stage runs the steps that hold pixels of non-palindromic bank rows a second time with the true coefficients (round 6; before: a plain
16-lane redo after the accept test); the shipped highres banks exercise that with 2 (8-bit) of 864 rows only (the 10-bit bank's 44 rows are
above the threshold: eight-load stage).  Here the highres bank gets 1 .. 400 of its rows perturbed (one tap
moved by an ulp, or a row replaced by noise), so that steps with 0, 1 and 4 such pixels, tail re-hash columns (width 134) and
Randomness blending all meet the redo path -- compared bit for bit with the CPU oracle on the same bank, and with the eight-load
stage (RAISR_HIP_SYM=0).  Above 16 such rows (RAISR_HIP_SYM_MAX_ROWS) the library keeps the eight-load stage by itself: same
comparison, other code path; RAISR_HIP_SYM_MAX_ROWS=1000 forces the redo path on those banks too.
(Raisr_AVX512.cpp:134-149, Raisr.cpp:1147-1200.)"""
import os

import numpy as np
import pytest

from common import folder, dtype_for

pytestmark = pytest.mark.gpu

FOLD = "filters_2x/filters_highres"


def _perturbed_bank(bits, nasym, seed):
    import raisr_hip as R
    bank, qstr, qcoh, qa = R.read_model_folder(folder(FOLD), bits, 1)
    rng = np.random.default_rng(seed)
    hk, pt, _ = bank.shape
    flat = bank.reshape(hk * pt, 121)
    rows = rng.choice(hk * pt, size=nasym, replace=False)
    for i, r in enumerate(rows):
        if i % 3 == 2:                                   # a row that is nowhere near a palindrome
            flat[r] = (rng.standard_normal(121) * 0.02).astype(np.float32)
            flat[r, 60] += np.float32(1.0)
        else:                                            # one tap off by an ulp, as in the shipped banks
            k = int(rng.integers(0, 121))
            if k == 60:
                k = 59
            flat[r, k] = np.nextafter(flat[r, k], np.float32(10.0))
    return bank, qstr, qcoh, qa


def _oracle(y, bank, bits, asm, blending, preset):
    import oracle_py as O
    m = O.Model(folder(FOLD), bits, 1)
    m.bank = np.ascontiguousarray(bank, np.float32)
    p = O.make_pass(m, bits, False, asm, blending)
    h, w = y.shape
    lr = O.resize(y, 2 * w, 2 * h)
    return O.run_pass(lr, p, preset=preset).astype(dtype_for(bits))


def _gpu(y, bank, qstr, qcoh, qa, bits, asm, blending, preset, env=None):
    import raisr_hip as R
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        dev = R.RaisrDevice(0)                            # the A/B switches are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    h, w = y.shape
    dev.set_model(0, bank, qstr, qcoh, qa)
    dev.configure(w, h, 2 * w, 2 * h, bits=bits, passes=1, mode=1, hash_variant=asm, blending=blending)
    out = preset.copy()
    dev.process_host(np.ascontiguousarray(y), out)
    dev.close()
    return out


@pytest.mark.parametrize("nasym", [1, 5, 16, 60, 170, 400])
@pytest.mark.parametrize("bits", [8, 10])
def test_non_palindromic_rows_bit_exact(nasym, bits):
    import oracle_py as O
    import raisr_hip as R
    import synth
    bank, qstr, qcoh, qa = _perturbed_bank(bits, nasym, seed=1000 + nasym)
    for (w, h) in ((96, 64), (134, 50)):
        for name in ("natural", "random"):
            y = synth.FRAME_KINDS[name](w, h, bits)
            for blending in (O.BLEND_COUNT, O.BLEND_RANDOMNESS):
                preset = np.full((2 * h, 2 * w), 77, dtype_for(bits))
                ref = _oracle(y, bank, bits, 2, blending, preset)
                got = _gpu(y, bank, qstr, qcoh, qa, bits, R.HASH_AVX512, blending, preset)
                assert np.array_equal(ref, got), (nasym, bits, w, h, name, blending, int((ref != got).sum()))
                plain = _gpu(y, bank, qstr, qcoh, qa, bits, R.HASH_AVX512, blending, preset, env={"RAISR_HIP_SYM": "0"})
                assert np.array_equal(plain, got), (nasym, bits, w, h, name, blending, "eight-load stage differs")
                forced = _gpu(y, bank, qstr, qcoh, qa, bits, R.HASH_AVX512, blending, preset, env={"RAISR_HIP_SYM_MAX_ROWS": "1000"})
                assert np.array_equal(forced, got), (nasym, bits, w, h, name, blending, "symmetric stage with second runs differs")



def test_redo_path_on_a_larger_frame_with_every_step_pattern():
    """640 x 360 -> 1280 x 720 random frame, 60 perturbed rows, symmetric stage forced: thousands of steps with 1, 2, 3 and 4 redone pixels."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    bank, qstr, qcoh, qa = _perturbed_bank(8, 60, seed=7)
    y = synth.random_y(640, 360, 8, seed=99)
    preset = np.zeros((720, 1280), np.uint8)
    ref = _oracle(y, bank, 8, 2, O.BLEND_COUNT, preset)
    got = _gpu(y, bank, qstr, qcoh, qa, 8, R.HASH_AVX512, O.BLEND_COUNT, preset, env={"RAISR_HIP_SYM_MAX_ROWS": "1000"})
    assert np.array_equal(ref, got), int((ref != got).sum())
