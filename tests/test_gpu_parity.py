"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, bit-exact, on seeded frames."""
import numpy as np
import pytest

from common import CASES, folder, dtype_for

pytestmark = pytest.mark.gpu


def _frames(w, h, bits):
    import synth
    return {
        "natural": synth.natural_y(w, h, bits, seed=12345),
        "random": synth.random_y(w, h, bits, seed=777),
        "checker": synth.checker_y(w, h, bits),
        "constant": synth.constant_y(w, h, bits),
    }


def _oracle(y, case):
    from common import oracle_y
    return oracle_y(y, case)


def _gpu(y, case):
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    dev = R.RaisrDevice(0)
    dev.set_model_from_folder(folder(fold), bits, passes)
    dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
    out = np.zeros((oh, ow), dtype_for(bits))
    dev.process_host(np.ascontiguousarray(y), out)
    stages = dev.read_stage(passes - 1)
    dev.close()
    return out, stages


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("size", [(96, 64), (134, 50), (200, 40)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_y_bit_exact(case, size):
    w, h = size
    bits = case[3]
    for name, y in _frames(w, h, bits).items():
        ref = _oracle(y, case)
        got, _ = _gpu(y, case)
        bad = np.argwhere(ref != got)
        assert bad.size == 0, f"{case[0]} {name} {w}x{h}: {len(bad)} mismatching pixels, first at {bad[:5].tolist()}"
