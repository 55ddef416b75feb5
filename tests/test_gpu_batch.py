"""-m gpu: frame batches (raisr_hip_process_y_device_batch): n frames through one launch per kernel give, frame for frame, the bits
of the oracle -- for every case of the matrix (fp32 and binary16 numerics, one and two passes, both two-pass modes, 10-bit),
n in {1, 3, 8}, equally spaced planes (one launch) and scattered ones (frame by frame)."""
import numpy as np
import pytest

from common import CASES, folder, dtype_for, oracle_y

pytestmark = pytest.mark.gpu


def _run(case, frames, scattered=False, twice=False):
    import torch
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = frames[0].shape
    ow, oh = w * rn // rd, h * rn // rd
    n = len(frames)
    tdt = torch.uint8 if bits == 8 else torch.uint16
    host = np.stack(frames)
    d_in = torch.from_numpy(host.view(np.int16) if bits != 8 else host).cuda().view(tdt)
    gap = 3 if scattered else 1                                       # scattered: planes of one allocation, but not equally spaced
    d_out = torch.zeros((n * gap + 1, oh, ow), dtype=tdt, device="cuda")
    slots = [i * gap + (1 if scattered and i == n - 1 else 0) for i in range(n)]
    bps = 1 if bits == 8 else 2
    dev = R.RaisrDevice(0)
    try:
        dev.set_model_from_folder(folder(fold), bits, passes)
        dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
        ins = [d_in[i].data_ptr() for i in range(n)]
        outs = [d_out[s].data_ptr() for s in slots]
        for _ in range(2 if twice else 1):
            dev.process_y_batch(ins, w * bps, outs, ow * bps)
        dev.synchronize()
        torch.cuda.synchronize()
        got = d_out.cpu().view(torch.int16 if bits != 8 else torch.uint8).numpy().view(dtype_for(bits))
    finally:
        dev.close()
    return [got[s] for s in slots]


def _frames(w, h, bits, n):
    import synth
    kinds = ["natural", "random", "checker", "constant"]
    out = []
    for i in range(n):
        k = kinds[i % 4]
        out.append(synth.natural_y(w, h, bits, seed=100 + i) if k == "natural" else
                   synth.random_y(w, h, bits, seed=200 + i) if k == "random" else synth.FRAME_KINDS[k](w, h, bits))
    return out


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("n", [1, 3, 8])
def test_batch_bit_exact(case, n):
    frames = _frames(134, 50, case[3], n)
    got = _run(case, frames, twice=(n == 3))
    for i, y in enumerate(frames):
        ref = oracle_y(y, case)
        bad = np.argwhere(ref != got[i])
        assert bad.size == 0, f"{case[0]} n={n} frame {i}: {len(bad)} mismatching pixels, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[13]], ids=lambda c: c[0])
def test_scattered_planes_run_frame_by_frame(case):
    frames = _frames(96, 64, case[3], 4)
    got = _run(case, frames, scattered=True)
    for i, y in enumerate(frames):
        assert np.array_equal(oracle_y(y, case), got[i]), (case[0], i)


def test_batch_then_single_then_larger_batch():
    """the scratch planes grow with the largest batch seen; single frames keep working in between"""
    import torch
    import raisr_hip as R
    case = CASES[0]
    w, h = 96, 64
    frames = _frames(w, h, 8, 6)
    refs = [oracle_y(y, case) for y in frames]
    d_in = torch.from_numpy(np.stack(frames)).cuda()
    d_out = torch.zeros((6, 2 * h, 2 * w), dtype=torch.uint8, device="cuda")
    dev = R.RaisrDevice(0)
    try:
        dev.set_model_from_folder(folder(case[1]), 8, 1)
        dev.configure(w, h, 2 * w, 2 * h, bits=8, passes=1, hash_variant=case[6])
        ins = [d_in[i].data_ptr() for i in range(6)]
        outs = [d_out[i].data_ptr() for i in range(6)]
        dev.process_y_batch(ins[:2], w, outs[:2], 2 * w)
        dev.process_y(ins[2], w, outs[2], 2 * w)
        dev.process_y_batch(ins[3:], w, outs[3:], 2 * w)
        dev.synchronize()
        got = d_out.cpu().numpy()
    finally:
        dev.close()
    for i in range(6):
        assert np.array_equal(refs[i], got[i]), i
