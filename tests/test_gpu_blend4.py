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


@pytest.mark.parametrize("bits", [8, 10, 16])
def test_resize_3_2_blocks_bit_exact(bits):
    """k_resize3x2 (a thread: 12 x 3 outputs) against the oracle's cheap upscale: sizes off the 12-column / 3-row block seams, both tie
    rules, 4-byte aligned and unaligned destination rows (12-sample stores / single-sample stores), extreme samples."""
    import oracle_py as O
    import raisr_hip as R
    import torch
    dt = dtype_for(bits)
    tdt = torch.uint8 if bits == 8 else torch.uint16
    bps = 1 if bits == 8 else 2
    rng = np.random.default_rng(bits)
    for tie in (R.TIE_HALF_UP, R.TIE_HALF_EVEN):
        dev = R.RaisrDevice(0)
        dev.set_model_from_folder(folder("filters_1.5x/filters_highres"), 8, 1)
        dev.configure(16, 16, 24, 24, bits=8, passes=1, mode=1, tie=tie)        # the plane entry takes the context's tie rule
        for sw, sh in ((16, 8), (22, 10), (86, 50), (128, 30), (130, 34), (1280, 6)):
            dw, dh = sw * 3 // 2, sh * 3 // 2
            planes = [rng.integers(0, 1 << bits, (sh, sw)).astype(dt), np.full((sh, sw), (1 << bits) - 1, dt), (np.indices((sh, sw)).sum(0) % 2 * ((1 << bits) - 1)).astype(dt)]
            for y in planes:
                ref = O.resize(y, dw, dh, tie).astype(dt)
                for pad in (0, 1, 4):                                            # destination pitch = dw + pad samples
                    d_in = torch.from_numpy(y).cuda()
                    d_out = torch.zeros((dh, dw + pad), dtype=tdt, device="cuda")
                    dev.resize_plane(d_in.data_ptr(), sw, sh, sw * bps, d_out.data_ptr(), dw, dh, (dw + pad) * bps, bits)
                    dev.synchronize()
                    got = d_out.cpu().numpy()
                    assert np.array_equal(got[:, :dw], ref), (bits, tie, sw, sh, pad, int((got[:, :dw] != ref).sum()))
                    assert not got[:, dw:].any()
        dev.close()


def test_blend_families_agree_on_random_geometries():
    """Seeded sweep: the four-columns-per-lane kernels (default) against the LDS-tile kernels (RAISR_HIP_BLEND_ROWS=0) on 36 random
    geometries whose output width is a multiple of 4 (8 ... 2 100 columns, 14 ... 260 rows), 8 / 10 bit, fp32 and binary16 numerics,
    one and two passes -- the whole output plane must agree (both families are separately pinned to the oracle above)."""
    import synth
    rng = np.random.default_rng(20260930)
    variants = [("filters_2x/filters_highres", (2, 1), 8, 1, 1, 2), ("filters_2x/filters_highres", (2, 1), 10, 1, 1, 2),
                ("filters_2x/filters_highres", (2, 1), 8, 1, 1, 5), ("filters_2x/filters_denoise", (2, 1), 8, 2, 2, 2),
                ("filters_1.5x/filters_highres", (3, 2), 8, 1, 1, 2), ("filters_2x/filters_lowres", (2, 1), 8, 2, 1, 1)]
    for i in range(36):
        fold, (rn, rd), bits, passes, mode, asm = variants[i % len(variants)]
        if rn == 2:
            w = 2 * int(rng.integers(2, 526))               # output width 8 ... 2 100, a multiple of 4
            h = int(rng.integers(7, 131))
        else:
            w = 8 * int(rng.integers(1, 175))               # 1.5x: output width 12 ... 2 088, a multiple of 12
            h = 2 * int(rng.integers(5, 87))
        case = ("x", fold, (rn, rd), bits, passes, mode, asm, bool(i & 1))
        y = synth.random_y(w, h, bits, seed=1000 + i) if i % 3 else synth.natural_y(w, h, bits, seed=1000 + i)
        a = _gpu(y, case, 0)
        b = _gpu(y, case, 8)
        assert np.array_equal(a, b), (i, fold, w, h, bits, passes, asm, int((a != b).sum()))
