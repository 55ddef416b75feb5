"""-m gpu: seeded fuzz over the PLUGIN host path (RNLHandler_Process / Submit+Collect): frame geometry around the row-range
thresholds of the last pass (out heights from no range to many tile rows), row paddings, bit depth, numerics flavour, passes,
pageable numpy planes vs planes from RNLHandler_HostAlloc (and mixed), yuv420 chroma -- Y against the oracle bit for bit, chroma
against the oracle's cheap upscale, padding bytes untouched."""
import os

import numpy as np
import pytest

from common import folder, oracle_y

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        w = int(rng.integers(12, 230)) * 2; h = int(rng.integers(10, 160)) * 2          # even: yuv420
        bits = 10 if rng.random() < 0.3 else 8
        asm = int(rng.choice([1, 2, 6])) if bits == 10 else int(rng.choice([1, 2, 5, 6]))
        passes = int(rng.choice([1, 2]))
        mode = int(rng.choice([1, 2])) if passes == 2 else 1
        fold = "filters_2x/filters_denoise" if mode == 2 else str(rng.choice(["filters_2x/filters_highres", "filters_2x/filters_lowres"]))
        suffix = "_2" if passes == 2 else ""
        if not os.path.exists(os.path.join(folder(fold), f"filterbin_2_{bits}{suffix}")):
            continue
        mem = str(rng.choice(["numpy", "hostalloc", "mixed"]))
        pads = [int(rng.choice([0, 0, 6, 32, 100])) for _ in range(6)]
        entry = str(rng.choice(["process", "process", "submit"]))
        out.append((w, h, bits, asm, passes, mode, fold, mem, tuple(pads), entry, int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("case", _cases(int(os.environ.get("RAISR_HOST_FUZZ_N", "48")), int(os.environ.get("RAISR_HOST_FUZZ_SEED", "20260929"))),
                         ids=lambda c: f"{c[0]}x{c[1]}_{c[2]}b_a{c[3]}_p{c[4]}m{c[5]}_{c[6].split('_')[-1]}_{c[7]}_{c[9]}")
def test_host_fuzz_case(case):
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h, bits, asm, passes, mode, fold, mem, pads, entry, seed = case
    dt = np.uint8 if bits == 8 else np.uint16
    cw, ch = w // 2, h // 2
    keep = []

    def plane(shape, pad, locked, fill=None):
        if locked:
            hp = R.HostPlane(shape, dt, shape[1] + pad)
            keep.append(hp)
            base = hp.array.base.reshape(shape[0], shape[1] + pad)     # the whole allocation, padding included
        else:
            base = np.zeros((shape[0], shape[1] + pad), dt)
        base[...] = 0xA5 if bits == 8 else 0x02A5
        a = base[:, :shape[1]]
        a[...] = 0 if fill is None else fill
        return base, a
    locked = {"numpy": [False] * 6, "hostalloc": [True] * 6, "mixed": [bool((seed >> i) & 1) for i in range(6)]}[mem]
    y = synth.natural_y(w, h, bits, seed=seed)
    u = synth.random_y(cw, ch, bits, seed=seed ^ 1).astype(dt); v = synth.random_y(cw, ch, bits, seed=seed ^ 2).astype(dt)
    shapes = [(h, w), (ch, cw), (ch, cw), (2 * h, 2 * w), (2 * ch, 2 * cw), (2 * ch, 2 * cw)]
    fills = [y, u, v, None, None, None]
    bases, views = zip(*[plane(shapes[i], pads[i], locked[i], fills[i]) for i in range(6)])
    ref_asm = 2 if asm == 6 else asm
    ref = oracle_y(y, ("x", fold, (2, 1), bits, passes, mode, ref_asm, False))
    ru = O.resize(u, 2 * cw, 2 * ch).astype(dt); rv = O.resize(v, 2 * cw, 2 * ch).astype(dt)
    pad_val = 0xA5 if bits == 8 else 0x02A5
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, bits, R.VideoRange, 20, asm, passes, mode) == 0
    try:
        assert R.RNLHandler_SetRes(views[:3], views[3:]) == 0
        for rep in range(2):
            for o in views[3:]:
                o[...] = 0
            if entry == "process":
                assert R.RNLHandler_Process(views[:3], views[3:]) == 0
            else:
                assert R.RNLHandler_SetAsyncDepth(2) == 0
                assert R.RNLHandler_Submit(views[:3], views[3:]) == 0
                assert R.RNLHandler_Collect() == 0
                assert R.RNLHandler_SetAsyncDepth(0) == 0
            bad = np.argwhere(views[3] != ref)
            assert bad.size == 0, (rep, len(bad), bad[:5].tolist())
            assert np.array_equal(views[4], ru) and np.array_equal(views[5], rv)
            for i in range(6):
                if pads[i]:
                    assert (bases[i][:, shapes[i][1]:] == pad_val).all(), ("padding touched", i)
    finally:
        assert R.RNLHandler_Deinit() == 0
        for hp in keep:
            hp.close()
