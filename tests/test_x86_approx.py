"""Oracle's software models of VRCP14/VRSQRT14/RCPPS/RSQRTPS against values captured from a real
GenuineIntel AVX-512 core (tests/golden/x86_approx_special.txt, written by
oracle/tools/capture_x86_approx.c) and, when this host IS such a CPU, against the hardware
exhaustively (all 2^23 mantissas x parities via the sha256 recorded in x86_approx_tables.h)."""
import ctypes
import hashlib
import os
import re
import struct
import subprocess
import numpy as np
import pytest

from common import ROOT


def _f(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def _u(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def _same(a, b):
    # NaN payloads must match too
    return a == b


def test_models_match_captured_hardware_values():
    import oracle_py as O
    L = O.lib()
    rows = [[int(t, 16) for t in ln.split()] for ln in open(os.path.join(ROOT, "tests/golden/x86_approx_special.txt"))]
    assert len(rows) > 3000
    bad = []
    for x, r14, s14, r, s in rows:
        xf = ctypes.c_float(_f(x))
        got = (_u(L.ora_x86_rcp14(xf)), _u(L.ora_x86_rsqrt14(xf)), _u(L.ora_x86_rcp(xf)), _u(L.ora_x86_rsqrt(xf)))
        if got != (r14, s14, r, s):
            bad.append((hex(x), [hex(g) for g in got], [hex(v) for v in (r14, s14, r, s)]))
    assert not bad, bad[:10]


def _is_intel_avx512():
    try:
        info = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return "GenuineIntel" in info and " avx512f" in info and " avx512vl" in info


@pytest.mark.skipif(not _is_intel_avx512(), reason="needs a GenuineIntel AVX-512 host (the instructions are vendor specific)")
def test_models_exhaustive_against_this_cpu(tmp_path):
    """Re-capture all six exhaustive dumps on this CPU and check (a) their sha256 equals the one the
    committed tables were fitted from and (b) the vectorised numpy restatement of the model
    formulas reproduces every one of the 6 x 2^23 results."""
    exe = tmp_path / "cap"
    subprocess.check_call(["gcc", "-O2", "-mavx512f", "-mavx512vl", os.path.join(ROOT, "oracle/tools/capture_x86_approx.c"), "-o", str(exe)])
    subprocess.check_call([str(exe), str(tmp_path)])
    hdr = open(os.path.join(ROOT, "oracle/x86_approx_tables.h")).read()
    want = re.search(r'X86_APPROX_DUMP_SHA256 "([0-9a-f]+)"', hdr).group(1)
    sha = hashlib.sha256()
    for n in ("rcp14", "rsqrt14_e0", "rsqrt14_e1", "rcp", "rsqrt_e0", "rsqrt_e1"):
        sha.update(open(tmp_path / f"{n}.bin", "rb").read())
    assert sha.hexdigest() == want

    def table(name, ctype):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", hdr, re.S).group(1)
        return np.array([int(t.rstrip("u")) for t in body.replace("\n", " ").split(",") if t.strip()], dtype=ctype)

    m = np.arange(1 << 23, dtype=np.int64)
    c0, c1 = table("X86_RCP14_C0", np.int64), table("X86_RCP14_C1", np.int64)
    code = (c0[m >> 17] - c1[m >> 17] * ((m >> 7) & 1023)) >> 9
    bits = (126 << 23) | (code << 7)
    bits[0] = 0x3F800000
    assert np.array_equal(bits.astype(np.uint32), np.fromfile(tmp_path / "rcp14.bin", dtype="<u4"))
    s0, s1 = table("X86_RSQRT14_C0", np.int64), table("X86_RSQRT14_C1", np.int64)
    for p, n in enumerate(("rsqrt14_e0", "rsqrt14_e1")):
        i = 32 * p + (m >> 18)
        code = (s0[i] - s1[i] * ((m >> 8) & 1023)) >> 9
        bits = (126 << 23) | (code << 7)
        if p == 0:
            bits[0] = 0x3F800000
        assert np.array_equal(bits.astype(np.uint32), np.fromfile(tmp_path / f"{n}.bin", dtype="<u4"))
    lut = table("X86_RCP_LUT", np.int64)
    assert np.array_equal(((126 << 23) | (lut[m >> 12] << 11)).astype(np.uint32), np.fromfile(tmp_path / "rcp.bin", dtype="<u4"))
    lut = table("X86_RSQRT_LUT", np.int64)
    for p, n in enumerate(("rsqrt_e0", "rsqrt_e1")):
        assert np.array_equal(((126 << 23) | (lut[1024 * p + (m >> 13)] << 11)).astype(np.uint32),
                              np.fromfile(tmp_path / f"{n}.bin", dtype="<u4"))


def test_product_and_oracle_tables_are_the_same_capture():
    a = open(os.path.join(ROOT, "oracle/x86_approx_tables.h")).read()
    b = open(os.path.join(ROOT, "video-super-resolution-library_amd/csrc/x86_approx_tables.h")).read()
    assert a == b
