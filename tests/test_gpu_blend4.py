"""-m gpu: the census blend's two kernel families against the oracle.  Planes whose rows are 4-sample aligned take k_blend4 /
k_blend4_16 (a lane owns four columns, a wave 256 columns x 4 / 8 / 16 rows, neighbours through DPP wave shifts: kernels_blend.h);
every other geometry, and RAISR_HIP_BLEND_ROWS=0, the 64 x 16 LDS-tile kernels.  Sizes straddle the 256 / 512 / 1024-column
wave and workgroup seams, heights the 4 / 8 / 16-row ones; 8- and 10-bit, binary16 pipeline, two passes, 1.5x."""
import os
import numpy as np
import pytest

from common import folder, dtype_for, oracle_y

pytestmark = pytest.mark.gpu

CASES4 = [
    ("2x_highres_8b", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False),
    ("2x_highres_10b", "filters_2x/filters_highres", (2, 1), 10, 1, 1, 2, False),
    ("2x_lowres_8b_avx2_full", "filters_2x/filters_lowres", (2, 1), 8, 1, 1, 1, True),
    ("2x_denoise_8b_2p_m2", "filters_2x/filters_denoise", (2, 1), 8, 2, 2, 2, False),
    ("2x_highres_8b_fp16", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 5, False),
    ("1.5x_denoise_8b_2p_m2_fp16", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 5, False),
    ("1.5x_highres_8b", "filters_1.5x/filters_highres", (3, 2), 8, 1, 1, 2, False),
]
# LR sizes: output widths 264 (one wave + 8 columns), 520 (two waves + 8), 1032 (one 16-row workgroup of four waves + 8), 256 exactly;
# output heights off the 4- / 8- / 16-row seams
SIZES = {(2, 1): [(132, 37), (260, 21), (516, 18), (128, 9)], (3, 2): [(176, 50), (344, 14)]}


def _gpu(y, case, rows):
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    old = os.environ.get("RAISR_HIP_BLEND_ROWS")
    os.environ["RAISR_HIP_BLEND_ROWS"] = str(rows)
    try:
        dev = R.RaisrDevice(0)
        dev.set_model_from_folder(folder(fold), bits, passes)
        dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
        out = np.full((oh, ow), 0x55, dtype_for(bits))
        dev.process_host(np.ascontiguousarray(y), out)
        dev.close()
    finally:
        if old is None:
            del os.environ["RAISR_HIP_BLEND_ROWS"]
        else:
            os.environ["RAISR_HIP_BLEND_ROWS"] = old
    return out


@pytest.mark.parametrize("case", CASES4, ids=[c[0] for c in CASES4])
def test_blend_kernels_bit_exact(case):
    import synth
    bits = case[3]
    for w, h in SIZES[case[2]]:
        for kind, y in (("natural", synth.natural_y(w, h, bits, seed=w + h)), ("random", synth.random_y(w, h, bits, seed=3 * w + h))):
            ref = oracle_y(y, case)
            for rows in (0, 4, 8, 16):
                got = _gpu(y, case, rows)
                bad = np.argwhere(ref != got)
                assert bad.size == 0, f"{case[0]} {kind} {w}x{h} rows={rows}: {len(bad)} mismatching pixels, first at {bad[:5].tolist()}"


def test_blend_extreme_samples():
    """Frames of the extreme sample values and 1-px patterns (census counts 0 and 8, clamps at both limits), every kernel family."""
    import synth
    case = CASES4[0]
    w, h = 260, 21
    frames = [np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8), synth.checker_y(w, h, 8),
              (np.indices((h, w)).sum(0) % 2 * 255).astype(np.uint8)]
    for y in frames:
        ref = oracle_y(y, case)
        for rows in (0, 4, 8, 16):
            assert np.array_equal(ref, _gpu(y, case, rows)), rows
