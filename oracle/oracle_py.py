"""oracle_py.py -- TEST INFRASTRUCTURE: ctypes front-end of oracle/libraisr_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.  It carries its own (numpy) reader of the trained-data folders so
that the oracle does not share the product's C++ model loader
(format: reference Library/Raisr.cpp:246-433,1531-1578).
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ASM_AVX2, ASM_AVX512, ASM_FP16 = 1, 2, 5
BLEND_RANDOMNESS, BLEND_COUNT = 1, 2
TIE_HALF_UP, TIE_HALF_EVEN = 0, 1


class OraPass(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_int), ("lo", ctypes.c_int), ("hi", ctypes.c_int),
                ("pixel_types", ctypes.c_int), ("asm_type", ctypes.c_int), ("blending", ctypes.c_int),
                ("qangle", ctypes.c_float), ("qstr", ctypes.c_float * 2), ("qcoh", ctypes.c_float * 2),
                ("bank", ctypes.c_void_p)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_ISA = "x86-64-v3 (AVX2)"


def _host_has_avx512():
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except (OSError, StopIteration):
        return False
    return all(f in flags for f in ("avx512f", "avx512vl", "avx512dq", "avx512bw"))


def isa():
    """Vector ISA the loaded oracle build was compiled for (reported by bench.py's cpu_baseline)."""
    lib()
    return _ISA


def lib():
    global _LIB, _ISA
    if _LIB is None:
        so = os.environ.get("RAISR_ORACLE_SO")                 # override: compiler cross-check
        if not so:
            # RAISR_ORACLE_ISA: "avx2" (default; what the parity tests use everywhere), "avx512", or "auto" = the
            # AVX-512 build of the same sources when this host executes it (bench.py's CPU baseline)
            want = os.environ.get("RAISR_ORACLE_ISA", "avx2")
            if want == "avx512" or (want == "auto" and _host_has_avx512()):
                so = os.path.join(_HERE, "libraisr_oracle_avx512.so")
                _ISA = "x86-64-v4 (AVX-512, 512-bit vectors)"
            else:
                so = os.path.join(_HERE, "libraisr_oracle.so")
        else:
            _ISA = "custom build " + os.path.basename(so)
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        for n in ("ora_x86_rcp14", "ora_x86_rsqrt14", "ora_x86_rcp", "ora_x86_rsqrt"):
            getattr(_LIB, n).restype = ctypes.c_float
            getattr(_LIB, n).argtypes = [ctypes.c_float]
        _LIB.ora_hash.restype = ctypes.c_int
        _LIB.ora_hash.argtypes = [ctypes.c_float] * 3 + [ctypes.c_void_p, ctypes.c_int]
        _LIB.ora_hash_array.restype = None
        _LIB.ora_hash_array.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return _LIB


def clamp_range(bits, full_range):
    """Library/Raisr.cpp:1451-1468"""
    if bits == 8:
        return (0, 255) if full_range else (16, 235)
    if bits == 10:
        return (0, 1023) if full_range else (64, 940)
    return (0, 65535)


class Model:
    """One pass worth of trained data read from a filter folder."""

    def __init__(self, folder, bits, pass_no):
        sfx = f"_2_{bits}" + ("_2" if pass_no == 2 else "")
        raw = open(os.path.join(folder, "filterbin" + sfx), "rb").read()
        assert raw[:4] == b"fp32", raw[:4]
        hk, pt, rows = np.frombuffer(raw[4:16], dtype="<u4")
        assert len(raw) - 16 == hk * pt * rows * 4
        self.hashkeys, self.pixel_types, self.rows = int(hk), int(pt), int(rows)
        self.bank = np.frombuffer(raw[16:], dtype="<f4").reshape(hk, pt, rows).copy()
        # std::stod then (float), Raisr.cpp:371-377,407-413
        self.qstr = [np.float32(float(t)) for t in open(os.path.join(folder, "Qfactor_strbin" + sfx)).read().split()]
        self.qcoh = [np.float32(float(t)) for t in open(os.path.join(folder, "Qfactor_cohbin" + sfx)).read().split()]
        cfg = open(os.path.join(folder, "config")).readline().split()
        self.qa, self.qs, self.qc, self.patch = (int(t) for t in cfg[:4])
        assert len(self.qstr) == 2 and len(self.qcoh) == 2 and self.patch == 11


def make_pass(model, bits, full_range=False, asm=ASM_AVX512, blending=BLEND_COUNT):
    lo, hi = clamp_range(bits, full_range)
    p = OraPass()
    p.bits, p.lo, p.hi = bits, lo, hi
    p.pixel_types = model.pixel_types
    p.asm_type, p.blending = asm, blending
    # gQAngle = gQuantizationAngle / PI with PI the float 3.141592653 (Raisr.cpp:1553, Raisr_globals.h:29)
    p.qangle = np.float32(np.float32(model.qa) / np.float32(3.141592653))
    p.qstr[0], p.qstr[1] = model.qstr
    p.qcoh[0], p.qcoh[1] = model.qcoh
    p._bank_keepalive = np.ascontiguousarray(model.bank, dtype=np.float32)
    p.bank = p._bank_keepalive.ctypes.data
    return p


def _u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


_LIB512 = None


def lib512():
    """The AVX-512 build (it alone carries the hand-vectorised twin raisr_oracle_avx512.c), or None where the host cannot run it."""
    global _LIB512
    if _LIB512 is None and _host_has_avx512():
        so = os.path.join(_HERE, "libraisr_oracle_avx512.so")
        if not os.path.exists(so):
            build()
        _LIB512 = ctypes.CDLL(so)
    return _LIB512


def process_y_intrinsics(plane, out_w, out_h, p1, p2=None, passes=1, mode=1, tie=TIE_HALF_UP):
    """process_y through ora512_process_y (own AVX-512 intrinsics; same bits as process_y by construction and by test)."""
    L = lib512()
    if L is None:
        raise RuntimeError("this host does not execute AVX-512")
    src = _u16(plane)
    h, w = src.shape
    out = np.zeros((out_h, out_w), dtype=np.uint16)
    L.ora512_process_y(src.ctypes.data_as(ctypes.c_void_p), w, h, out.ctypes.data_as(ctypes.c_void_p), out_w, out_h,
                       passes, mode, ctypes.byref(p1), ctypes.byref(p2 if p2 is not None else p1), tie)
    return out


def resize(plane, out_w, out_h, tie=TIE_HALF_UP):
    src = _u16(plane)
    h, w = src.shape
    dst = np.empty((out_h, out_w), dtype=np.uint16)
    lib().ora_resize_bilinear(src.ctypes.data_as(ctypes.c_void_p), w, h, w,
                              dst.ctypes.data_as(ctypes.c_void_p), out_w, out_h, out_w, tie)
    return dst


def run_pass(lr, p, preset=None, dumps=False):
    lr = _u16(lr)
    h, w = lr.shape
    out = np.zeros((h, w), dtype=np.uint16) if preset is None else _u16(preset).copy()
    hd = np.empty((h, w), dtype=np.int32) if dumps else None
    hr = np.empty((h, w), dtype=np.float32) if dumps else None
    lib().ora_pass(lr.ctypes.data_as(ctypes.c_void_p), w, h, ctypes.byref(p),
                   out.ctypes.data_as(ctypes.c_void_p),
                   hd.ctypes.data_as(ctypes.c_void_p) if dumps else None,
                   hr.ctypes.data_as(ctypes.c_void_p) if dumps else None)
    return (out, hd, hr) if dumps else out


def process_y(plane, out_w, out_h, p1, p2=None, passes=1, mode=1, tie=TIE_HALF_UP):
    src = _u16(plane)
    h, w = src.shape
    out = np.zeros((out_h, out_w), dtype=np.uint16)
    lib().ora_process_y(src.ctypes.data_as(ctypes.c_void_p), w, h,
                        out.ctypes.data_as(ctypes.c_void_p), out_w, out_h,
                        passes, mode, ctypes.byref(p1), ctypes.byref(p2 if p2 is not None else p1), tie)
    return out


def upscale_y(plane, folder, ratio=2.0, bits=8, full_range=False, passes=1, mode=1,
              asm=ASM_AVX512, blending=BLEND_COUNT, tie=TIE_HALF_UP):
    """Convenience: whole reference-equivalent Y job from a filter folder."""
    h, w = plane.shape
    out_w, out_h = int(w * ratio), int(h * ratio)
    m1 = Model(folder, bits, 1)
    p1 = make_pass(m1, bits, full_range, asm, blending)
    p2 = None
    if passes == 2:
        p2 = make_pass(Model(folder, bits, 2), bits, full_range, asm, blending)
    return process_y(plane, out_w, out_h, p1, p2, passes, mode, tie)


# ------------------------------------------------------------------------------------------------
# AVX512-FP16 path (binary16 arithmetic, 8-bit content)
# ------------------------------------------------------------------------------------------------
class OraPass16(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_int), ("lo", ctypes.c_int), ("hi", ctypes.c_int),
                ("pixel_types", ctypes.c_int), ("blending", ctypes.c_int),
                ("qangle", ctypes.c_uint16), ("qstr", ctypes.c_uint16 * 2), ("qcoh", ctypes.c_uint16 * 2),
                ("bank", ctypes.c_void_p)]


def _h(x):
    """python float (double) -> binary16 bit pattern, single RNE rounding"""
    return int(np.array([x], dtype=np.float64).astype(np.float16).view(np.uint16)[0])


def make_pass16(folder, bits, pass_no, full_range=False, blending=BLEND_COUNT):
    """Model conversion of ReadTrainedData<_Float16> (Raisr.cpp:344-350,375,411): weights
    (fp16)(float), thresholds (fp16)stod(token), gQAngle (float) -> fp16."""
    m = Model(folder, bits, pass_no)
    lo, hi = clamp_range(bits, full_range)
    p = OraPass16()
    p.bits, p.lo, p.hi = bits, lo, hi
    p.pixel_types, p.blending = m.pixel_types, blending
    qangle32 = np.float32(np.float32(m.qa) / np.float32(3.141592653))
    p.qangle = int(np.array([qangle32]).astype(np.float16).view(np.uint16)[0])
    sfx = f"_2_{bits}" + ("_2" if pass_no == 2 else "")
    qs = [float(t) for t in open(os.path.join(folder, "Qfactor_strbin" + sfx)).read().split()]
    qc = [float(t) for t in open(os.path.join(folder, "Qfactor_cohbin" + sfx)).read().split()]
    p.qstr[0], p.qstr[1] = _h(qs[0]), _h(qs[1])
    p.qcoh[0], p.qcoh[1] = _h(qc[0]), _h(qc[1])
    p._bank_keepalive = np.ascontiguousarray(m.bank.astype(np.float16).view(np.uint16))
    p.bank = p._bank_keepalive.ctypes.data
    return p


def run_pass16(lr, p, dumps=False, preset=None):
    lr = _u16(lr)
    h, w = lr.shape
    out = np.zeros((h, w), dtype=np.uint16) if preset is None else _u16(preset).copy()
    hd = np.empty((h, w), dtype=np.int32) if dumps else None
    hr = np.empty((h, w), dtype=np.uint16) if dumps else None
    lib().ora16_pass(lr.ctypes.data_as(ctypes.c_void_p), w, h, ctypes.byref(p), out.ctypes.data_as(ctypes.c_void_p),
                     hd.ctypes.data_as(ctypes.c_void_p) if dumps else None,
                     hr.ctypes.data_as(ctypes.c_void_p) if dumps else None)
    return (out, hd, hr) if dumps else out


def process_y16(plane, out_w, out_h, p1, p2=None, passes=1, mode=1, tie=TIE_HALF_UP):
    src = _u16(plane)
    h, w = src.shape
    out = np.zeros((out_h, out_w), dtype=np.uint16)
    lib().ora16_process_y(src.ctypes.data_as(ctypes.c_void_p), w, h, out.ctypes.data_as(ctypes.c_void_p), out_w, out_h,
                          passes, mode, ctypes.byref(p1), ctypes.byref(p2 if p2 is not None else p1), tie)
    return out


def hash_array(abd, p, avx2_variant):
    """Hash buckets of an (n, 3) float32 array of (a, b, d) triples (oracle hash_pixel)."""
    abd = np.ascontiguousarray(abd, np.float32)
    out = np.zeros(abd.shape[0], np.uint8)
    lib().ora_hash_array(abd.ctypes.data_as(ctypes.c_void_p), abd.shape[0], ctypes.byref(p), int(avx2_variant),
                         out.ctypes.data_as(ctypes.c_void_p))
    return out
