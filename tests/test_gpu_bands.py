"""-m gpu: frames cut into horizontal bands (independent sub-frames, include/raisr_hip.h "Horizontal bands") give
the whole-frame result bit for bit -- through RNLHandler_Process (RAISR_HIP_BANDS forces the cut on small test
frames) for every pipeline flavour, both blending modes and yuv420 chroma."""
import os

import numpy as np
import pytest

from common import folder, oracle_y

pytestmark = pytest.mark.gpu

# (folder, ratio, (num, den), bits, asm, passes, mode)
FLAVOURS = [
    ("filters_2x/filters_highres", 2.0, (2, 1), 8, 2, 1, 1),
    ("filters_2x/filters_lowres", 2.0, (2, 1), 8, 1, 1, 1),
    ("filters_2x/filters_highres", 2.0, (2, 1), 8, 2, 2, 1),
    ("filters_2x/filters_denoise", 2.0, (2, 1), 8, 2, 2, 2),
    ("filters_2x/filters_highres", 2.0, (2, 1), 10, 2, 1, 1),
    ("filters_1.5x/filters_highres", 1.5, (3, 2), 8, 2, 1, 1),
    ("filters_1.5x/filters_denoise", 1.5, (3, 2), 8, 5, 2, 2),
    ("filters_2x/filters_highres", 2.0, (2, 1), 8, 5, 2, 1),
]


@pytest.mark.parametrize("flavour", FLAVOURS, ids=lambda f: f"{f[0].split('/')[1]}_{f[1]}x_{f[3]}b_asm{f[4]}_{f[5]}p_m{f[6]}")
@pytest.mark.parametrize("nbands", [2, 3])
def test_banded_rnlprocess_equals_whole_frame(flavour, nbands, monkeypatch):
    import oracle_py as O
    import raisr_hip as R
    import synth
    fold, ratio, (rn, rd), bits, asm, passes, mode = flavour
    w, h = 112, 160
    dt = np.uint8 if bits == 8 else np.uint16
    y = synth.random_y(w, h, bits, seed=31 + nbands) if passes == 1 else synth.natural_y(w, h, bits, seed=32 + nbands)
    u = synth.random_y(w // 2, h // 2, bits, seed=8).astype(dt)
    v = synth.random_y(w // 2, h // 2, bits, seed=9).astype(dt)
    ow, oh = w * rn // rd, h * rn // rd
    oy = np.zeros((oh, ow), dt); ou = np.zeros((oh // 2, ow // 2), dt); ov = np.zeros((oh // 2, ow // 2), dt)
    assert len(R.plan_bands(h, oh, passes, nbands)) == nbands          # the test frame really is cut
    monkeypatch.setenv("RAISR_HIP_BANDS", str(nbands))
    assert R.RNLHandler_Init(folder(fold), ratio, bits, R.VideoRange, 20, asm, passes, mode) == 0
    try:
        assert R.RNLHandler_SetRes((y, u, v), (oy, ou, ov)) == 0
        assert R.RNLHandler_Process((y, u, v), (oy, ou, ov), R.CountOfBitsChanged) == 0
        ref = oracle_y(y, ("x", fold, (rn, rd), bits, passes, mode, asm, False))
        bad = np.argwhere(oy != ref)
        assert bad.size == 0, (len(bad), bad[:5])
        assert np.array_equal(ou, O.resize(u, ow // 2, oh // 2).astype(dt))
        assert np.array_equal(ov, O.resize(v, ow // 2, oh // 2).astype(dt))
    finally:
        assert R.RNLHandler_Deinit() == 0


@pytest.mark.parametrize("asm", [2, 5])
def test_banded_randomness_blending_keeps_the_unwritten_pixels(asm, monkeypatch):
    """blending=1: the pixels [c_final, W-6) of frame row H-7 are never written (reference quirk);
    with bands they must still be the caller's bytes, and every other pixel the whole-frame value."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h = 100, 128
    y = synth.random_y(w, h, 8, seed=77)
    c = synth.chroma(w // 2, h // 2, 8)
    outs = []
    for nb in (1, 3):
        monkeypatch.setenv("RAISR_HIP_BANDS", str(nb))
        oy = np.full((2 * h, 2 * w), 201, np.uint8); ou = np.zeros((h, w), np.uint8); ov = np.zeros((h, w), np.uint8)
        assert R.RNLHandler_Init(folder("filters_2x/filters_highres"), 2.0, 8, R.VideoRange, 20, asm, 1, 1) == 0
        try:
            assert R.RNLHandler_SetRes((y, c, c), (oy, ou, ov)) == 0
            assert R.RNLHandler_Process((y, c, c), (oy, ou, ov), R.Randomness) == 0
        finally:
            assert R.RNLHandler_Deinit() == 0
        outs.append(oy)
    assert np.array_equal(outs[0], outs[1])
    W, H = 2 * w, 2 * h
    c_final = 6 + 8 * ((W - 12) // 8)
    assert np.all(outs[1][H - 7, c_final:W - 6] == 201)


@pytest.mark.parametrize("passes,mode", [(1, 1), (2, 2)])
def test_one_frame_split_over_ranks_device_layer(passes, mode):
    """The multi-GPU latency recipe (sharding.band_for_rank): every "rank" -- here three contexts on one GPU --
    processes only its band's input rows with the ordinary device entry point; the kept rows stitched together are
    the whole-frame output."""
    import raisr_hip as R
    import sharding
    import synth
    import torch
    w, h, world = 128, 168, 3
    fold = "filters_2x/filters_highres" if mode == 1 else "filters_2x/filters_denoise"
    y = synth.natural_y(w, h, 8, seed=5)
    ref = oracle_y(y, ("x", fold, (2, 1), 8, passes, mode, 2, False))
    out = np.zeros_like(ref)
    for rank in range(world):
        b = sharding.band_for_rank(h, 2 * h, passes, rank, world)
        assert b is not None
        dev = R.RaisrDevice(0)
        try:
            dev.set_model_from_folder(folder(fold), 8, passes)
            dev.configure(w, b["in_row_count"], 2 * w, b["out_row_count"], bits=8, passes=passes, mode=mode)
            d_in = torch.from_numpy(np.ascontiguousarray(y[b["in_row_begin"]:b["in_row_begin"] + b["in_row_count"]])).cuda()
            d_out = torch.empty((b["out_row_count"], 2 * w), dtype=torch.uint8, device="cuda")
            dev.process_y(d_in.data_ptr(), w, d_out.data_ptr(), 2 * w)
            dev.synchronize()
            sub = d_out.cpu().numpy()
        finally:
            dev.close()
        k0 = b["keep_begin"] - b["out_row_begin"]
        out[b["keep_begin"]:b["keep_begin"] + b["keep_count"]] = sub[k0:k0 + b["keep_count"]]
    assert np.array_equal(out, ref)
