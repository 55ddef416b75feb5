"""-m gpu: seeded fuzz over geometry and options -- widths/heights that are not multiples of any tile or chunk size
(64-column tiles, 16-row tiles, 8/16/32-column hash chunks, 2- and 3-row upscale periods), every numerics flavour,
pass count / mode, range, bit depth and blending mode -- HIP path vs oracle, bit for bit."""
import os

import numpy as np
import pytest

from common import folder, dtype_for

pytestmark = pytest.mark.gpu


MAX_W = int(os.environ.get("RAISR_FUZZ_MAX_W", "150"))      # one-off wider sweeps: RAISR_FUZZ_N / _SEED / _MAX_W / _MAX_H
MAX_H = int(os.environ.get("RAISR_FUZZ_MAX_H", "110"))


def _has_model(fold, bits, passes):
    suffix = "_2" if passes == 2 else ""
    return os.path.exists(os.path.join(folder(fold), f"filterbin_2_{bits}{suffix}"))


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        ratio = (2, 1) if rng.random() < 0.65 else (3, 2)
        w = int(rng.integers(7, MAX_W)); h = int(rng.integers(7, MAX_H))
        if ratio == (3, 2):
            w -= w % 2; h -= h % 2                  # 1.5x: even input sizes give integer output sizes
        bits = 10 if (ratio == (2, 1) and rng.random() < 0.25) else 8
        asm = int(rng.choice([1, 2, 5])) if bits == 8 else int(rng.choice([1, 2]))
        passes = int(rng.choice([1, 2]))
        mode = int(rng.choice([1, 2])) if passes == 2 else 1
        kinds = ["highres", "lowres", "denoise"] if ratio == (2, 1) else ["highres", "denoise"]
        kind = str(rng.choice(kinds))
        base = "filters_2x" if ratio == (2, 1) else "filters_1.5x"
        if not _has_model(f"{base}/filters_{kind}", bits, passes):
            continue                                # e.g. 1.5x highres ships no second-pass bank
        full = bool(rng.random() < 0.3)
        blending = 1 if (passes == 1 and rng.random() < 0.3) else 2     # Randomness: one pass (see test_randomness_blending_bit_exact)
        frame = str(rng.choice(["random", "natural", "checker"]))
        out.append((f"{base}/filters_{kind}", ratio, w, h, bits, asm, passes, mode, full, blending, frame, int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("case", _cases(int(os.environ.get("RAISR_FUZZ_N", "240")), int(os.environ.get("RAISR_FUZZ_SEED", "20260928"))), ids=lambda c: f"{c[0].split('/')[1]}_{c[1][0]}-{c[1][1]}_{c[2]}x{c[3]}_{c[4]}b_a{c[5]}_p{c[6]}m{c[7]}_{'f' if c[8] else 'v'}_b{c[9]}_{c[10]}")
def test_fuzz_case(case):
    import oracle_py as O
    import raisr_hip as R
    import synth
    fold, (rn, rd), w, h, bits, asm, passes, mode, full, blending, frame, seed = case
    ow, oh = w * rn // rd, h * rn // rd
    y = {"random": lambda: synth.random_y(w, h, bits, seed=seed), "natural": lambda: synth.natural_y(w, h, bits, seed=seed),
         "checker": lambda: synth.checker_y(w, h, bits)}[frame]()
    # oracle (Randomness never writes a few pixels: both sides start from the same preset plane)
    preset = np.full((oh, ow), 77, np.uint16)
    if asm == 5:
        p1 = O.make_pass16(folder(fold), bits, 1, full, blending)
        p2 = O.make_pass16(folder(fold), bits, 2, full, blending) if passes == 2 else None
        ref = O.run_pass16(O.resize(y, ow, oh), p1, preset=preset) if blending == 1 else O.process_y16(y, ow, oh, p1, p2, passes, mode)
    else:
        p1 = O.make_pass(O.Model(folder(fold), bits, 1), bits, full, asm, blending)
        p2 = O.make_pass(O.Model(folder(fold), bits, 2), bits, full, asm, blending) if passes == 2 else None
        ref = O.run_pass(O.resize(y, ow, oh), p1, preset=preset) if blending == 1 else O.process_y(y, ow, oh, p1, p2, passes, mode)
    ref = ref.astype(dtype_for(bits))
    # HIP path
    dev = R.RaisrDevice(0)
    try:
        dev.set_model_from_folder(folder(fold), bits, passes)
        dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm, blending=blending)
        out = np.full((oh, ow), 77, dtype_for(bits))
        dev.process_host(np.ascontiguousarray(y), out)
    finally:
        dev.close()
    bad = np.argwhere(out != ref)
    assert bad.size == 0, (len(bad), bad[:5].tolist())
