"""-m gpu: the HIP path against the committed fixtures of tests/golden/ -- a comparison in which the oracle does not run.

oracle_digests.json (28 small frames over the whole case matrix), baseline_digests.json (the five BASELINE.json configurations at
full size: SURVEY s8(c)(iii)) and the per-stage dumps of one case (s8(c)(ii): cheap upscale, bucket per pixel, HR plane bits,
output).  Self-generated regression pins (tests/make_golden.py), not evidence about the reference."""
import hashlib
import json
import os

import numpy as np
import pytest

from common import CASES, folder, dtype_for
from make_golden import BASELINE, baseline_frame, row_checksums

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _hip(y, case, stages=False):
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    dev = R.RaisrDevice(0, hooks=stages)                    # per-stage dumps: the test-hooks flavour (include/raisr_hip_debug.h)
    try:
        dev.set_model_from_folder(folder(fold), bits, passes)
        dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
        out = np.zeros((oh, ow), dtype_for(bits))
        if stages:
            dev.keep_stages(True)
        dev.process_host(np.ascontiguousarray(y), out)
        st = dev.read_stage(passes - 1) if stages else None
    finally:
        dev.close()
    return out, st


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_small_frames_match_the_committed_digests(case):
    import synth
    want = json.load(open(os.path.join(GOLD, "oracle_digests.json")))
    bits = case[3]
    for nm, fr in (("natural", synth.natural_y(96, 64, bits, seed=4242)), ("random", synth.random_y(96, 64, bits, seed=99))):
        out, _ = _hip(fr, case)
        assert hashlib.sha256(out.tobytes()).hexdigest() == want[f"{case[0]}/{nm}"], (case[0], nm)


@pytest.mark.parametrize("name", sorted(BASELINE))
def test_baseline_configurations_match_the_committed_digests_at_full_size(name):
    want = json.load(open(os.path.join(GOLD, "baseline_digests.json")))[name]
    y = baseline_frame(name)
    assert hashlib.sha256(y.tobytes()).hexdigest() == want["input_sha256"]
    out, _ = _hip(y, BASELINE[name][0])
    if hashlib.sha256(out.tobytes()).hexdigest() != want["sha256"]:
        rows = [64 * i for i, (a, b) in enumerate(zip(row_checksums(out), want["row_adler32_every_64"])) if a != b]
        raise AssertionError(f"{name}: HIP output differs from the committed digest; sampled rows that differ: {rows[:10]}")


def test_stage_dumps():
    z = np.load(os.path.join(GOLD, "stages_2x_highres_8b_96x64.npz"))
    out, (hs, hr) = _hip(z["input"], CASES[0], stages=True)
    assert np.array_equal(out, z["out"])
    zone = z["hash"] != 0xFF                                   # the kernels write hash / HR inside the filtered zone only
    assert np.array_equal(hs[zone], z["hash"][zone])
    assert np.array_equal(np.ascontiguousarray(hr, np.float32).view(np.uint32)[zone], z["hr_bits"][zone])
