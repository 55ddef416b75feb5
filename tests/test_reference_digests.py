"""Oracle (CPU) and HIP path (-m gpu) against digests of the REAL reference -- when tests/golden/reference_digests.json exists.

That file can only be produced on a host with Intel IPP and an Intel AVX-512 CPU (tools/pin_against_reference/): it holds sha256 of
the unmodified reference's Y output for the case matrix and the five BASELINE configurations, and which tie rule of the cheap
upscale reproduces IPP.  In a checkout without it these tests SKIP, loudly: parity with the reference is then unpinned
 -- a property of the build environment, not of the code."""
import hashlib
import json
import os

import numpy as np
import pytest

from common import CASES, folder, dtype_for, oracle_y
from make_golden import BASELINE, baseline_frame

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_digests.json")
WHY = ("PARITY UNPINNED: tests/golden/reference_digests.json is absent -- run tools/pin_against_reference/ on a host with Intel IPP "
       "(oneAPI 2021.12 / 2022.0) and an Intel AVX-512 CPU, commit the file, and these tests compare the oracle and the HIP path with the real reference")


def _load():
    if not os.path.exists(PATH):
        pytest.skip(WHY)
    d = json.load(open(PATH))
    tie = {"half_up": 0, "half_even": 1}.get(d["tie_rule_that_reproduces_ipp"])
    return d, tie


def _jobs():
    import synth
    for case in CASES:
        bits = case[3]
        for nm, fr in (("natural", lambda b=bits: synth.natural_y(96, 64, b, seed=4242)), ("random", lambda b=bits: synth.random_y(96, 64, b, seed=99))):
            yield f"{case[0]}/{nm}", case, fr
    for name in sorted(BASELINE):
        yield f"{name}/full", BASELINE[name][0], (lambda n=name: baseline_frame(n))


def test_the_kit_found_a_tie_rule():
    d, tie = _load()
    assert tie is not None, ("neither tie rule of the cheap upscale reproduces IPP on every case: the build-defined stage differs from "
                             f"ippiResizeLinear ({d['cases_bit_exact']} of {d['cases_run']} cases bit-exact) -- see the per-case counts in the file")


@pytest.mark.parametrize("jid", [j[0] for j in _jobs()])
def test_oracle_equals_the_reference(jid):
    d, tie = _load()
    rec = d["cases"].get(jid)
    if rec is None or "skipped" in rec:
        pytest.skip(f"reference did not run {jid}: {rec and rec['skipped']}")
    if tie is None:
        pytest.skip("no tie rule reproduces IPP (reported by test_the_kit_found_a_tie_rule)")
    _, case, frame = next(j for j in _jobs() if j[0] == jid)
    y = frame()
    assert hashlib.sha256(y.tobytes()).hexdigest() == rec["in_sha256"], "the frame generator drifted since the digests were made"
    out = oracle_y(y, case, tie)
    assert hashlib.sha256(out.tobytes()).hexdigest() == rec["strict_sha256"], jid


@pytest.mark.gpu
@pytest.mark.parametrize("jid", [j[0] for j in _jobs()])
def test_hip_equals_the_reference(jid):
    import raisr_hip as R
    d, tie = _load()
    rec = d["cases"].get(jid)
    if rec is None or "skipped" in rec or tie is None:
        pytest.skip(f"no reference digest to compare {jid} with")
    _, case, frame = next(j for j in _jobs() if j[0] == jid)
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    y = frame()
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    dev = R.RaisrDevice(0)
    try:
        dev.set_model_from_folder(folder(fold), bits, passes)
        dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm, tie=tie)
        out = np.zeros((oh, ow), dtype_for(bits))
        dev.process_host(np.ascontiguousarray(y), out)
    finally:
        dev.close()
    assert hashlib.sha256(out.tobytes()).hexdigest() == rec["strict_sha256"], jid
