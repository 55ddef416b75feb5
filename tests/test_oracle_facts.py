"""The oracle against the known-answer facts SURVEY.md s8(a5,a6,a8,a13,c) records from the compiled
reference (the reference ships no golden vectors; these are its observed behaviours), and against
the regression fixtures under tests/golden/ (self-generated: see tests/golden/README.md)."""
import ctypes
import hashlib
import json
import os
import numpy as np
import pytest

from common import ROOT, folder, CASES, dtype_for, oracle_y


def _p(fold="filters_2x/filters_highres", bits=8, asm=2, full=False, pass_no=1):
    import oracle_py as O
    return O.make_pass(O.Model(folder(fold), bits, pass_no), bits, full, asm)


def test_flat_patch_hashes_to_bucket_207():
    # a=b=d=0 => angle ~ pi - 1.8e-6 => angleIdx 23, strength/coherence idx 0
    import oracle_py as O
    p = _p()
    assert O.lib().ora_hash(0.0, 0.0, 0.0, ctypes.byref(p), 0) == 207
    assert O.lib().ora_hash(0.0, 0.0, 0.0, ctypes.byref(p), 1) == 207


def test_constant_frame_is_identity_after_clamp():
    import oracle_py as O, synth
    out = O.upscale_y(synth.constant_y(40, 30, 8, 128), folder("filters_2x/filters_highres"))
    assert out.shape == (60, 80) and np.all(out == 128)
    out = O.upscale_y(synth.constant_y(40, 30, 8, 250), folder("filters_2x/filters_highres"))
    assert np.all(out[1:-1, 1:-1] == 235) and np.all(out[0] == 250) and np.all(out[:, 0] == 250)   # borders keep unclamped LR


@pytest.mark.parametrize("W,asm,first,last,tail", [
    (3840, 2, 6, 3829, (3822, 3829)), (1920, 2, 6, 1909, (1902, 1909)), (7680, 2, 6, 7669, (7662, 7669)),
    (1920, 1, 6, 1909, None)])
def test_column_coverage_rule(W, asm, first, last, tail):
    """Filtered columns are [6, c_final); in AVX-512 mode the last 8 of them are re-hashed by the
    AVX2 routine.  Probed with a frame whose hash dump marks filtered pixels."""
    import oracle_py as O
    rng = np.random.default_rng(1)
    lr = rng.integers(0, 256, (14, W)).astype(np.uint16)
    p = _p(asm=asm)
    _, hd, _ = O.run_pass(lr, p, dumps=True)
    cols = np.nonzero(hd[6] >= 0)[0]
    assert cols[0] == first and cols[-1] == last and len(cols) == last - first + 1
    assert np.all(hd[:6] < 0) and np.all(hd[8:] < 0)


def test_border_policy():
    """row 0 / H-1 and col 0 / W-1 keep the unclamped LR; rows 1..5, H-6..H-2, cols 1..5 and
    [c_final, W-1) are clip(LR)."""
    import oracle_py as O, synth
    lr = synth.random_y(70, 40, 8).astype(np.uint16)
    out = O.run_pass(lr, _p())
    clip = np.clip(lr, 16, 235)
    assert np.array_equal(out[0], lr[0]) and np.array_equal(out[-1], lr[-1])
    assert np.array_equal(out[:, 0], lr[:, 0]) and np.array_equal(out[:, -1], lr[:, -1])
    assert np.array_equal(out[1:6, 1:-1], clip[1:6, 1:-1]) and np.array_equal(out[-6:-1, 1:-1], clip[-6:-1, 1:-1])
    assert np.array_equal(out[1:-1, 1:6], clip[1:-1, 1:6])
    c_final = 6 + 8 * ((70 - 12) // 8)
    assert np.array_equal(out[1:-1, c_final:-1], clip[1:-1, c_final:-1])


def test_bilinear_2x_is_9331_kernel():
    import oracle_py as O, synth
    src = synth.random_y(33, 21, 8).astype(np.int64)
    got = O.resize(src, 66, 42).astype(np.int64)
    pad = np.pad(src, 1, mode="edge")
    for dy in (0, 1):
        for dx in (0, 1):
            # output (2i+dy, 2j+dx): near tap weight 3, far tap weight 1, far tap is i-1 for dy=0 else i+1
            ny = pad[1:-1] if True else None
            near_r = pad[1:-1]; far_r = pad[0:-2] if dy == 0 else pad[2:]
            def hmix(rows):
                near_c = rows[:, 1:-1]; far_c = rows[:, 0:-2] if dx == 0 else rows[:, 2:]
                return 3 * near_c + far_c
            want = (3 * hmix(near_r) + hmix(far_r) + 8) >> 4
            assert np.array_equal(got[dy::2, dx::2], want)


def test_two_pass_mode1_equals_manual_composition():
    import oracle_py as O, synth
    y = synth.natural_y(48, 36, 8)
    p1, p2 = _p(), _p(pass_no=2)
    a = O.process_y(y, 96, 72, p1, p2, 2, 1)
    lr = O.resize(y, 96, 72)
    b = O.run_pass(O.run_pass(lr, p1), p2)
    assert np.array_equal(a, b)


def test_regression_fixtures():
    """sha256 of the oracle output for every CASE on seeded frames (drift detector for the oracle;
    the -m gpu suite checks the HIP path against the same digests)."""
    import oracle_py as O, synth
    want = json.load(open(os.path.join(ROOT, "tests/golden/oracle_digests.json")))
    got = {}
    for cid, fold, (rn, rd), bits, passes, mode, asm, full in CASES:
        y = synth.natural_y(96, 64, bits, seed=4242) if bits == 8 else synth.natural_y(96, 64, bits, seed=4242)
        r = synth.random_y(96, 64, bits, seed=99)
        for nm, fr in (("natural", y), ("random", r)):
            out = oracle_y(fr, (cid, fold, (rn, rd), bits, passes, mode, asm, full))
            got[f"{cid}/{nm}"] = hashlib.sha256(out.tobytes()).hexdigest()
    assert got == want


def test_vectorised_gtwg_row_equals_the_scalar_form():
    """The CPU-baseline-friendly loop order (gtwg_row) must give the bits of the line-by-line
    restatement (gtwg_pixel) for every pixel parity and bit depth."""
    import oracle_py as O
    L = O.lib()
    L.ora_gtwg_both.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(11)
    for bits in (8, 10):
        plane = rng.integers(0, 1 << bits, (40, 64)).astype(np.float32)
        out = np.zeros(6, np.float32)
        for r in (6, 17, 33):
            for c in (6, 7, 20, 41, 57):
                L.ora_gtwg_both(plane.ctypes.data, 64, r, c, bits, out.ctypes.data)
                assert np.array_equal(out[:3].view(np.uint32), out[3:].view(np.uint32)), (bits, r, c, out)


def test_oracle_bits_do_not_depend_on_the_compiler(tmp_path):
    """The oracle's contract is one IEEE operation per source operation, so another compiler at another optimisation
    level, without AVX2/FMA code generation and without OpenMP, must give the same digests: builds the oracle with
    clang -O1 for plain x86-64 and compares a few cases with the committed fixtures."""
    import shutil
    import subprocess
    import sys
    clang = shutil.which("clang") or "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no second compiler")
    so = tmp_path / "libraisr_oracle_alt.so"
    src = [os.path.join(ROOT, "oracle", f) for f in ("raisr_oracle.c", "raisr_oracle_fp16.c")]
    cc = [clang, "-O1", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-Wno-unknown-pragmas", "-shared", "-o", str(so)] + src + ["-lm"]
    out = subprocess.run(cc, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    code = r'''
import hashlib, json, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, "video-super-resolution-library_amd")
from common import CASES, oracle_y
import synth
got = {}
for case in CASES:
    if case[0] not in ("2x_highres_8b_1p_avx512", "2x_lowres_8b_1p_avx2", "1.5x_denoise_8b_2p_m2_fp16", "2x_highres_10b_2p_m1_full"):
        continue
    for nm, fr in (("natural", synth.natural_y(96, 64, case[3], seed=4242)), ("random", synth.random_y(96, 64, case[3], seed=99))):
        got[f"{case[0]}/{nm}"] = hashlib.sha256(oracle_y(fr, case).tobytes()).hexdigest()
print(json.dumps(got))
'''
    env = dict(os.environ, RAISR_ORACLE_SO=str(so), OMP_NUM_THREADS="1")
    run = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    got = json.loads(run.stdout.strip().splitlines()[-1])
    want = json.load(open(os.path.join(ROOT, "tests/golden/oracle_digests.json")))
    assert len(got) == 8
    for k, v in got.items():
        assert want[k] == v, k


def test_avx512_build_of_the_oracle_gives_the_same_digests():
    """bench.py's cpu_baseline uses the AVX-512 build of the oracle sources on hosts that execute it
    (oracle/Makefile: libraisr_oracle_avx512.so); it must reproduce the committed digests bit for bit."""
    import subprocess
    import sys
    import oracle_py as O
    if not O._host_has_avx512():
        pytest.skip("host does not execute AVX-512")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libraisr_oracle_avx512.so"])
    code = r'''
import hashlib, json, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, "video-super-resolution-library_amd")
from common import CASES, oracle_y
import synth, oracle_py
got = {}
for case in CASES:
    for nm, fr in (("natural", synth.natural_y(96, 64, case[3], seed=4242)), ("random", synth.random_y(96, 64, case[3], seed=99))):
        got[f"{case[0]}/{nm}"] = hashlib.sha256(oracle_y(fr, case).tobytes()).hexdigest()
assert "AVX-512" in oracle_py.isa(), oracle_py.isa()
print(json.dumps(got))
'''
    env = dict(os.environ, RAISR_ORACLE_ISA="avx512")
    env.pop("RAISR_ORACLE_SO", None)
    run = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    got = json.loads(run.stdout.strip().splitlines()[-1])
    want = json.load(open(os.path.join(ROOT, "tests/golden/oracle_digests.json")))
    assert len(got) == 2 * len(CASES)
    for k, v in got.items():
        assert want[k] == v, k


def test_resize_fast_path_equals_the_generic_form():
    """ora_resize_bilinear reduces the ratios and works in 32 bits with a multiply-shift division where that is exact; the generic
    64-bit form is kept as the definition.  Random geometries, ratios and both tie rules."""
    import ctypes
    import oracle_py as O
    L = O.lib()
    rng = np.random.default_rng(11)
    for it in range(200):
        sw, sh = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        r = float(rng.choice([2.0, 1.5, 1.0, 1.7, 1.25, 1.99, 3.0]))
        dw, dh = max(1, int(sw * r)), max(1, int(sh * r))
        src = rng.integers(0, 65536, (sh, sw)).astype(np.uint16)
        for tie in (0, 1):
            a = np.zeros((dh, dw), np.uint16); b = np.zeros((dh, dw), np.uint16)
            L.ora_resize_bilinear(src.ctypes.data_as(ctypes.c_void_p), sw, sh, sw, a.ctypes.data_as(ctypes.c_void_p), dw, dh, dw, tie)
            L.ora_resize_bilinear_generic(src.ctypes.data_as(ctypes.c_void_p), sw, sh, sw, b.ctypes.data_as(ctypes.c_void_p), dw, dh, dw, tie)
            assert np.array_equal(a, b), (sw, sh, dw, dh, tie)
