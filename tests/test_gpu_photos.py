"""-m gpu: REAL picture content through the HIP path vs the CPU oracle, bit-exact.

The reference's only test assets are real clips (test/validation_suite/run_tests_avxout.sh:13,50-107; docs/performance.md:14).  None is
available offline; the photographs installed with this image's Python packages are (photos.py: 21 sources x 5 variants -- as decoded,
JPEG-blocky, letterboxed, bicubically enlarged, mosaic).  Four of the five BASELINE configurations rest on the certified hash stage's
bounds: these tests put content nobody synthesised through them.

  * BASELINE C1..C5 and the reference's published configuration (C2b: 1080p->4K 10-bit, docs/performance.md:8-14) at FULL size,
    one frame per variant, sources rotating;
  * the 14 CASES of the parity matrix at 416x240 (the reference's validation clips are small too), every variant.
"""
import numpy as np
import pytest

from common import CASES, folder, dtype_for, oracle_y

pytestmark = pytest.mark.gpu

# id, folder, ratio, bits, passes, mode, asm, full_range, (in_w, in_h)
FULL = [
    ("C1_540p_lowres_avx2", "filters_2x/filters_lowres", (2, 1), 8, 1, 1, 1, False, (960, 540)),
    ("C2_1080p_highres_1p", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False, (1920, 1080)),
    ("C2b_1080p_highres_1p_10bit", "filters_2x/filters_highres", (2, 1), 10, 1, 1, 2, False, (1920, 1080)),
    ("C3_1080p_highres_2p", "filters_2x/filters_highres", (2, 1), 8, 2, 1, 2, False, (1920, 1080)),
    ("C4_720p_1.5x_denoise_fp16_2p_m2", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 5, False, (1280, 720)),
    ("C5_4k_8k_10bit", "filters_2x/filters_highres", (2, 1), 10, 1, 1, 2, False, (3840, 2160)),
]


def _photos():
    import photos
    if len(photos.available()) < 3:
        pytest.skip("fewer than three photographs installed in this image")
    return photos


def _gpu_many(frames, case):
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case[:8]
    h, w = frames[0].shape
    ow, oh = w * rn // rd, h * rn // rd
    dev = R.RaisrDevice(0)
    dev.set_model_from_folder(folder(fold), bits, passes)
    dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
    outs = []
    for y in frames:
        out = np.zeros((oh, ow), dtype_for(bits))
        dev.process_host(np.ascontiguousarray(y), out)
        outs.append(out)
    dev.close()
    return outs


def _indices(k, salt):
    """one frame index per variant, the source rotating with `salt` (photos.photo_y: index % n = source, index // n = variant)"""
    import photos
    n = len(photos.available())
    return [v * n + (salt * 5 + 3 * v + k) % n for v in range(len(photos.VARIANTS))]


@pytest.mark.parametrize("case", FULL, ids=[c[0] for c in FULL])
def test_photographs_full_size_bit_exact(case):
    P = _photos()
    w, h = case[8]
    bits = case[3]
    idx = _indices(0, FULL.index(case))
    if w >= 3840:
        idx = idx[:3] + idx[4:]                     # soft2x of an 8K job adds nothing the 4K jobs do not cover; keeps the oracle's time down
    frames = P.frames(w, h, bits, idx)
    outs = _gpu_many(frames, case)
    for i, y, got in zip(idx, frames, outs):
        ref = oracle_y(y, case[:8])
        bad = np.argwhere(ref != got)
        assert bad.size == 0, (f"{case[0]} photo index {i} ({P.available()[i % len(P.available())]}, {P.VARIANTS[i // len(P.available()) % 5]}): "
                               f"{len(bad)} mismatching pixels, first {bad[:5].tolist()}")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_photographs_case_matrix_416x240(case):
    P = _photos()
    bits = case[3]
    idx = _indices(1, CASES.index(case)) + _indices(2, CASES.index(case) + 7)[:2]
    frames = P.frames(416, 240, bits, idx)
    outs = _gpu_many(frames, case)
    for i, y, got in zip(idx, frames, outs):
        ref = oracle_y(y, case)
        bad = np.argwhere(ref != got)
        assert bad.size == 0, f"{case[0]} photo index {i}: {len(bad)} mismatching pixels, first {bad[:5].tolist()}"
