"""ctypes bindings of libraisr_hip.so -- the MI355X Enhanced-RAISR library.

Two layers, both straight over the C ABI (no torch types cross the boundary):

* `RNLHandler_*` -- the reference's plugin API (reference Library/RaisrHandler.h:15-48) exactly as
  FFmpeg's vf_raisr calls it: host planes in, host planes out.  `upscale_frame_host()` is the
  call protocol of vf_raisr.c:146,286-312 (Init once, SetRes on the first frame, Process per frame).
* `RaisrDevice` -- the device-resident hot path (`raisr_hip_*`, include/raisr_hip.h) used by the
  benchmark and the parity tests: planes live in HBM (e.g. torch tensors' data_ptr()), launches go
  to a caller-supplied HIP stream.

The extension is REQUIRED: importing succeeds without it, but any use raises RuntimeError -- there
is no CPU fallback in the product.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAISR_HIP_LIB: an experimental build of the same library (scripts/build_exp.sh) for A/B runs on one GPU box; never a fallback
_SO = os.environ.get("RAISR_HIP_LIB") or os.path.join(_HERE, "libraisr_hip.so")
# the same sources with -DRAISR_HIP_TESTHOOKS: the product library plus the raisr_hip_debug_* entry points (include/raisr_hip_debug.h)
# and the selectable pipelines that exist for comparisons only (RAISR_HIP_SPLIT, RAISR_HIP_DEFER).  Tests and bench.py's self-check
# legs load it through RaisrDevice(..., hooks=True); the product library does not carry any of it.
_SO_HOOKS = os.environ.get("RAISR_HIP_LIB") or os.path.join(_HERE, "libraisr_hip_testhooks.so")
_LIB = None
_LIB_HOOKS = None

# RaisrDefaults.h enums
RNLErrorNone = 0
RNLErrorInsufficientResources = ctypes.c_int(0x80001000 - (1 << 32)).value
RNLErrorUndefined = ctypes.c_int(0x80001001 - (1 << 32)).value
RNLErrorBadParameter = ctypes.c_int(0x80001002 - (1 << 32)).value
Randomness, CountOfBitsChanged = 1, 2
AVX2, AVX512, OpenCL, OpenCLExternal, AVX512_FP16, HIP, HIPExternal = 1, 2, 3, 4, 5, 6, 7
VideoRange, FullRange = 1, 2

HASH_AVX2, HASH_AVX512, HASH_FP16 = 1, 2, 5
BLEND_RANDOMNESS, BLEND_COUNT = 1, 2
TIE_HALF_UP, TIE_HALF_EVEN = 0, 1


class VideoDataType(ctypes.Structure):
    _fields_ = [("pData", ctypes.c_void_p), ("width", ctypes.c_uint), ("height", ctypes.c_uint),
                ("step", ctypes.c_uint), ("bitShift", ctypes.c_uint)]


class RaisrHipBand(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("in_row_begin", "in_row_count", "out_row_begin", "out_row_count",
                                            "keep_begin", "keep_count")]


class RaisrHipRows(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("y_skip", "y_keep", "c_skip", "c_keep", "stage")]


class RaisrHipConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "in_width", "in_height", "out_width", "out_height", "bits", "clamp_lo", "clamp_hi", "passes",
        "two_pass_mode", "hash_variant", "blending", "use_pixel_type", "tie_rule")]


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(_SO) or not os.path.exists(_SO_HOOKS):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libraisr_hip.so", "libraisr_hip_testhooks.so"] + (["-B"] if force else []))
    return _SO


def _load(path):
    if True:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C {_HERE}` (hipcc, gfx950). "
                               "There is no CPU fallback.")
        # libraisr_hip.so and PyTorch-ROCm both need "libamdhip64.so.7"; the dynamic loader keeps one
        # copy per SONAME, so whichever is loaded first serves both.  PyTorch only works on its own
        # bundled runtime, hence load it first whenever it is installed (tensors' data_ptr()s and
        # streams are then valid in this library because there is exactly one HIP runtime).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(path)
        L.raisr_hip_last_error.restype = ctypes.c_char_p
        L.raisr_hip_version.restype = ctypes.c_char_p
        L.raisr_hip_model_blob_bytes.restype = ctypes.c_size_t
        L.raisr_hip_model_blob_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.raisr_hip_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
        L.raisr_hip_destroy.argtypes = [ctypes.c_void_p]
        L.raisr_hip_destroy.restype = None
        L.raisr_hip_set_model.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.raisr_hip_pack_model_blob.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.raisr_hip_set_model_blob_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.raisr_hip_configure.argtypes = [ctypes.c_void_p, ctypes.POINTER(RaisrHipConfig)]
        if os.environ.get("RAISR_HIP_LIB") is None or hasattr(L, "raisr_hip_process_y_device_batch"):
            L.raisr_hip_process_y_device_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                            ctypes.c_size_t, ctypes.c_void_p]
        L.raisr_hip_process_y_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                 ctypes.c_size_t, ctypes.c_void_p]
        L.raisr_hip_resize_plane_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        L.raisr_hip_process_frame_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                     ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.raisr_hip_process_host.argtypes = ([ctypes.c_void_p] + [ctypes.c_void_p, ctypes.c_size_t] * 6 + [ctypes.c_int] * 4)
        L.raisr_hip_synchronize.argtypes = [ctypes.c_void_p]
        L.raisr_hip_set_blending.argtypes = [ctypes.c_void_p, ctypes.c_int]
        if hasattr(L, "raisr_hip_debug_keep_stages"):
            L.raisr_hip_debug_keep_stages.argtypes = [ctypes.c_void_p, ctypes.c_int]
        if hasattr(L, "raisr_hip_debug_read_stage"):
            L.raisr_hip_debug_read_stage.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.raisr_hip_plan_bands.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(RaisrHipBand)]
        if hasattr(L, "raisr_hip_debug_hash"):
            L.raisr_hip_debug_hash.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.raisr_hip_stream_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]
        if os.environ.get("RAISR_HIP_LIB") is None or hasattr(L, "raisr_hip_stream_create_multi"):     # (an older A/B build may lack the multi-device ring)
            L.raisr_hip_stream_create_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            L.raisr_hip_stream_device_count.argtypes = [ctypes.c_void_p]
            L.raisr_hip_stream_device_of_frame.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
            L.raisr_hip_parse_device_list.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int]
            L.raisr_hip_parse_device_list_n.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
            L.raisr_hip_ring_slot.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
            L.raisr_hip_ring_slot.restype = None
            L.raisr_hip_broadcast_model_blob_devices.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
            L.RNLHandler_SetDeviceList.argtypes = [ctypes.c_char_p]
        L.raisr_hip_stream_destroy.argtypes = [ctypes.c_void_p]
        L.raisr_hip_stream_destroy.restype = None
        L.raisr_hip_stream_depth.argtypes = [ctypes.c_void_p]
        L.raisr_hip_stream_set_model_blob_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.raisr_hip_stream_set_fast.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.raisr_hip_stream_set_model.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.raisr_hip_stream_configure.argtypes = [ctypes.c_void_p, ctypes.POINTER(RaisrHipConfig)]
        L.raisr_hip_stream_submit.argtypes = ([ctypes.c_void_p] + [ctypes.c_void_p, ctypes.c_size_t] * 6 + [ctypes.c_int] * 4)
        L.raisr_hip_stream_collect.argtypes = [ctypes.c_void_p]
        L.raisr_hip_stream_in_flight.argtypes = [ctypes.c_void_p]
        L.raisr_hip_host_alloc.argtypes = [ctypes.c_size_t]
        L.raisr_hip_host_alloc.restype = ctypes.c_void_p
        L.raisr_hip_host_free.argtypes = [ctypes.c_void_p]
        L.raisr_hip_host_free.restype = None
        L.raisr_hip_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.raisr_hip_host_unregister.argtypes = [ctypes.c_void_p]
        L.raisr_hip_host_is_page_locked.argtypes = [ctypes.c_void_p]
        L.RNLHandler_HostAlloc.argtypes = [ctypes.c_size_t]
        L.RNLHandler_HostAlloc.restype = ctypes.c_void_p
        L.RNLHandler_HostFree.argtypes = [ctypes.c_void_p]
        L.RNLHandler_HostFree.restype = None
        L.raisr_hip_packed_frame_layout.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
        if hasattr(L, "raisr_hip_debug_read_c1tab"):
            L.raisr_hip_debug_read_c1tab.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        if hasattr(L, "raisr_hip_debug_approx_hash"):
            L.raisr_hip_debug_approx_hash.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        if hasattr(L, "raisr_hip_debug_fold16_check"):
            L.raisr_hip_debug_fold16_check.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        if hasattr(L, "raisr_hip_debug_certify"):
            L.raisr_hip_debug_certify.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.raisr_hip_set_fast.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.raisr_hip_use_streams.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.raisr_hip_broadcast_model_blob.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.raisr_hip_get_fast.argtypes = [ctypes.c_void_p]
        if hasattr(L, "raisr_hip_debug_certify_stats"):
            L.raisr_hip_debug_certify_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.raisr_hip_kernel_timing_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.raisr_hip_kernel_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.RNLHandler_Init.argtypes = [ctypes.c_char_p, ctypes.c_float, ctypes.c_uint, ctypes.c_int, ctypes.c_uint,
                                      ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
        vp = ctypes.POINTER(VideoDataType)
        L.RNLHandler_SetRes.argtypes = [vp] * 6
        L.RNLHandler_Process.argtypes = [vp] * 6 + [ctypes.c_int]
        L.RNLHandler_Submit.argtypes = [vp] * 6 + [ctypes.c_int]
        L.RNLHandler_SetAsyncDepth.argtypes = [ctypes.c_uint]
        L.raisr_hip_stream_set_blending.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.RNLHandler_SetOpenCLContext.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        return L




def lib():
    """The product library."""
    global _LIB
    if _LIB is None:
        _LIB = _load(_SO)
    return _LIB


def lib_hooks():
    """The test-hooks flavour (raisr_hip_debug_*, comparison pipelines)."""
    global _LIB_HOOKS
    if _LIB_HOOKS is None:
        _LIB_HOOKS = lib() if _SO_HOOKS == _SO else _load(_SO_HOOKS)
    return _LIB_HOOKS


def last_error():
    return lib().raisr_hip_last_error().decode()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {last_error()}")


def clamp_range(bits, full_range):
    """gMin/gMax of RNLInit (reference Library/Raisr.cpp:1451-1468)."""
    if bits == 8:
        return (0, 255) if full_range else (16, 235)
    if bits == 10:
        return (0, 1023) if full_range else (64, 940)
    return (0, 65535)


# ----------------------------------------------------------------------------------------------
# Reference plugin API (host planes)
# ----------------------------------------------------------------------------------------------
def plan_bands(in_height, out_height, passes, nbands):
    """Band plan of include/raisr_hip.h (pure host arithmetic, no GPU): list of dicts, one per band."""
    arr = (RaisrHipBand * max(1, nbands))()
    k = lib().raisr_hip_plan_bands(in_height, out_height, passes, nbands, arr)
    if k < 0:
        raise ValueError(last_error())
    return [{n: getattr(arr[i], n) for n, _ in RaisrHipBand._fields_} for i in range(k)]


def RNLHandler_Init(model_path, ratio, bit_depth=8, range_type=VideoRange, thread_count=20, asm_type=AVX512,
                    passes=1, two_pass_mode=1):
    return lib().RNLHandler_Init(os.fsencode(model_path), ratio, bit_depth, range_type, thread_count, asm_type,
                                 passes, two_pass_mode)


def RNLHandler_SetOpenCLContext(platform_index=0, device_index=0, stream=None):
    """device selection hook; `stream` (a hipStream_t value) only matters for asm = HIPExternal"""
    return lib().RNLHandler_SetOpenCLContext(stream, None, platform_index, device_index)


def RNLHandler_Deinit():
    return lib().RNLHandler_Deinit()


def RNLHandler_SetAsyncDepth(depth):
    return int(lib().RNLHandler_SetAsyncDepth(ctypes.c_uint(depth)))


def RNLHandler_AsyncCapacity():
    """Frames Submit accepts before a Collect is due: depth x the GPUs of the device list."""
    return lib().RNLHandler_AsyncCapacity()


def RNLHandler_SetDeviceList(devices):
    """GPUs of the asynchronous ring: "0,1,2" | "all" | "" (RaisrHandler.h)."""
    return lib().RNLHandler_SetDeviceList(devices.encode() if isinstance(devices, str) else devices)


def parse_device_list(text, devices_present=None):
    """-> list of device ordinals, or None for a malformed list (raisr_hip_parse_device_list[_n])."""
    buf = (ctypes.c_int * 16)()
    t = text.encode()
    n = lib().raisr_hip_parse_device_list(t, buf, 16) if devices_present is None else lib().raisr_hip_parse_device_list_n(t, devices_present, buf, 16)
    return None if n < 0 else [int(buf[i]) for i in range(n)]


def ring_slot(n_devices, depth, frame_index):
    """(device slot, lane on that device) of frame `frame_index` in an n-device ring (raisr_hip_ring_slot)."""
    d, l = ctypes.c_int(), ctypes.c_int()
    lib().raisr_hip_ring_slot(n_devices, depth, frame_index, ctypes.byref(d), ctypes.byref(l))
    return d.value, l.value


def RNLHandler_Submit(in_planes, out_planes, blending=CountOfBitsChanged):
    """Asynchronous counterpart of RNLHandler_Process: the numpy planes must stay alive and untouched until the matching Collect."""
    d = [_vdt(p) for p in list(in_planes) + list(out_planes)]
    return int(lib().RNLHandler_Submit(*[ctypes.byref(x) for x in d], blending))


def RNLHandler_Collect():
    return int(lib().RNLHandler_Collect())


def RNLHandler_FramesInFlight():
    return int(lib().RNLHandler_FramesInFlight())


class HostPlane:
    """A 2-D numpy array over page-locked memory from RNLHandler_HostAlloc (the plugin-level allocator a host's frame pool
    would use); `pitch_elems` > width leaves padding at the end of every row, as pool frames have."""

    def __init__(self, shape, dtype, pitch_elems=None):
        h, w = shape
        pitch = w if pitch_elems is None else int(pitch_elems)
        self.nbytes = h * pitch * np.dtype(dtype).itemsize
        self._p = lib().RNLHandler_HostAlloc(self.nbytes)
        if not self._p:
            raise MemoryError("RNLHandler_HostAlloc failed")
        buf = (ctypes.c_uint8 * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(h, pitch)[:, :w]

    def close(self):
        if self._p:
            self.array = None
            lib().RNLHandler_HostFree(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _vdt(a):
    v = VideoDataType()
    v.pData = a.ctypes.data
    v.height, v.width = a.shape
    v.step = a.strides[0]
    v.bitShift = 0
    return v


def RNLHandler_SetRes(in_planes, out_planes):
    d = [_vdt(p) for p in list(in_planes) + list(out_planes)]
    return lib().RNLHandler_SetRes(*[ctypes.byref(x) for x in d])


def RNLHandler_Process(in_planes, out_planes, blending=CountOfBitsChanged):
    d = [_vdt(p) for p in list(in_planes) + list(out_planes)]
    return lib().RNLHandler_Process(*[ctypes.byref(x) for x in d], blending)


def upscale_frame_host(y, u, v, model_path, ratio=2.0, bits=8, range_type=VideoRange, asm_type=AVX512, passes=1,
                       mode=1, blending=CountOfBitsChanged, device=0):
    """One frame through the RNLHandler call protocol of vf_raisr.c.  Planes are 2-D numpy arrays
    (uint8 for 8-bit, uint16 otherwise); returns (Y, U, V) output arrays."""
    oh, ow = int(y.shape[0] * ratio), int(y.shape[1] * ratio)
    och, ocw = int(u.shape[0] * ratio), int(u.shape[1] * ratio)
    oy = np.zeros((oh, ow), y.dtype); ou = np.zeros((och, ocw), u.dtype); ov = np.zeros((och, ocw), v.dtype)
    RNLHandler_SetOpenCLContext(0, device)
    rc = RNLHandler_Init(model_path, ratio, bits, range_type, 20, asm_type, passes, mode)
    if rc != RNLErrorNone:
        raise RuntimeError(f"RNLHandler_Init rc={rc:#x}")
    try:
        rc = RNLHandler_SetRes((y, u, v), (oy, ou, ov))
        if rc != RNLErrorNone:
            raise RuntimeError(f"RNLHandler_SetRes rc={rc:#x}")
        rc = RNLHandler_Process((y, u, v), (oy, ou, ov), blending)
        if rc != RNLErrorNone:
            raise RuntimeError(f"RNLHandler_Process rc={rc:#x}")
    finally:
        RNLHandler_Deinit()
    return oy, ou, ov


# ----------------------------------------------------------------------------------------------
# Trained-data folder -> arrays (host-side helper for the device API; the C++ loader in
# csrc/raisr_api.cpp is the one RNLHandler_Init uses)
# ----------------------------------------------------------------------------------------------
def read_model_folder(folder, bits, pass_no):
    sfx = f"_2_{bits}" + ("_2" if pass_no == 2 else "")
    raw = open(os.path.join(folder, "filterbin" + sfx), "rb").read()
    if raw[:4] != b"fp32":
        raise ValueError("hashtable corrupted")
    hk, pt, rows = (int(t) for t in np.frombuffer(raw[4:16], dtype="<u4"))
    if len(raw) - 16 != hk * pt * rows * 4 or rows != 121:
        raise ValueError("hashtable corrupted")
    bank = np.frombuffer(raw[16:], dtype="<f4").reshape(hk, pt, rows).copy()
    qstr = np.array([float(t) for t in open(os.path.join(folder, "Qfactor_strbin" + sfx)).read().split()], np.float64)
    qcoh = np.array([float(t) for t in open(os.path.join(folder, "Qfactor_cohbin" + sfx)).read().split()], np.float64)
    qa = int(open(os.path.join(folder, "config")).readline().split()[0])
    return bank, qstr, qcoh, qa


def pack_model_blob(bank, qstr, qcoh, quant_angle):
    """Device-layout model blob as a numpy uint8 array (what gets RCCL-broadcast between ranks)."""
    hk, pt, _ = bank.shape
    n = lib().raisr_hip_model_blob_bytes(hk, pt)
    blob = np.zeros(n, np.uint8)
    bank = np.ascontiguousarray(bank, np.float32)
    qstr = np.ascontiguousarray(qstr, np.float64); qcoh = np.ascontiguousarray(qcoh, np.float64)
    _check(lib().raisr_hip_pack_model_blob(blob.ctypes.data, bank.ctypes.data, hk, pt, qstr.ctypes.data, qcoh.ctypes.data,
                                           quant_angle), "pack_model_blob")
    return blob


class RaisrDevice:
    """One in-flight-frame lane of the device-resident hot path."""

    def __init__(self, device=0, hooks=False):
        """hooks=True: a context of the test-hooks flavour (libraisr_hip_testhooks.so): the debug_* / certify_* / keep_stages methods and the
        comparison pipelines (RAISR_HIP_SPLIT, RAISR_HIP_DEFER) exist there only."""
        self._L = lib_hooks() if hooks else lib()
        self._h = ctypes.c_void_p()
        self._check(self._L.raisr_hip_create(ctypes.byref(self._h), device), "raisr_hip_create")
        self.cfg = None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._L.raisr_hip_last_error().decode()}")

    def _hook(self, name):
        f = getattr(self._L, name, None)
        if f is None:
            raise RuntimeError(f"{name} is not part of the product library: create the device with hooks=True (libraisr_hip_testhooks.so)")
        return f

    def close(self):
        if self._h:
            self._L.raisr_hip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model_from_folder(self, folder, bits, passes=1):
        for p in range(passes):
            bank, qstr, qcoh, qa = read_model_folder(folder, bits, p + 1)
            self.set_model(p, bank, qstr, qcoh, qa)

    def set_model(self, pass_index, bank, qstr, qcoh, quant_angle):
        bank = np.ascontiguousarray(bank, np.float32)
        qstr = np.ascontiguousarray(qstr, np.float64); qcoh = np.ascontiguousarray(qcoh, np.float64)
        hk, pt, _ = bank.shape
        self._check(self._L.raisr_hip_set_model(self._h, pass_index, bank.ctypes.data, hk, pt, qstr.ctypes.data, qcoh.ctypes.data,
                                         quant_angle), "raisr_hip_set_model")

    def set_model_blob_device(self, pass_index, dev_ptr, nbytes, stream=None):
        self._check(self._L.raisr_hip_set_model_blob_device(self._h, pass_index, dev_ptr, nbytes, stream), "set_model_blob_device")

    def configure(self, in_w, in_h, out_w, out_h, bits=8, full_range=False, passes=1, mode=1, hash_variant=HASH_AVX512,
                  blending=BLEND_COUNT, ratio2=None, tie=TIE_HALF_UP):
        lo, hi = clamp_range(bits, full_range)
        c = RaisrHipConfig()
        c.in_width, c.in_height, c.out_width, c.out_height = in_w, in_h, out_w, out_h
        c.bits, c.clamp_lo, c.clamp_hi = bits, lo, hi
        c.passes, c.two_pass_mode = passes, mode
        c.hash_variant, c.blending = hash_variant, blending
        c.use_pixel_type = int(out_w == 2 * in_w and out_h == 2 * in_h) if ratio2 is None else int(ratio2)
        c.tie_rule = tie
        self._check(self._L.raisr_hip_configure(self._h, ctypes.byref(c)), "raisr_hip_configure")
        self.cfg = c

    def process_y(self, d_in, in_pitch, d_out, out_pitch, stream=None):
        self._check(self._L.raisr_hip_process_y_device(self._h, d_in, in_pitch, d_out, out_pitch, stream), "raisr_hip_process_y_device")

    def process_y_batch(self, d_in_list, in_pitch, d_out_list, out_pitch, stream=None):
        """n device-resident frames through one launch per kernel (equally spaced planes; otherwise frame by frame)"""
        n = len(d_in_list)
        key = (tuple(d_in_list), tuple(d_out_list))
        cache = self.__dict__.setdefault("_batch_tabs", {})
        if key not in cache:                                   # pointer tables are reused by loops that resubmit the same planes
            cache[key] = ((ctypes.c_void_p * n)(*d_in_list), (ctypes.c_void_p * n)(*d_out_list))
        a_in, a_out = cache[key]
        self._check(self._L.raisr_hip_process_y_device_batch(self._h, n, a_in, in_pitch, a_out, out_pitch, stream), "raisr_hip_process_y_device_batch")

    def resize_plane(self, d_src, sw, sh, spitch, d_dst, dw, dh, dpitch, bits, stream=None):
        self._check(self._L.raisr_hip_resize_plane_device(self._h, d_src, sw, sh, spitch, d_dst, dw, dh, dpitch, bits, stream),
               "raisr_hip_resize_plane_device")

    def process_frame(self, d_y, y_pitch, d_oy, oy_pitch, d_u, d_v, c_pitch, d_ou, d_ov, oc_pitch, cw, ch, ocw, och, stream=None):
        self._check(self._L.raisr_hip_process_frame_device(self._h, d_y, y_pitch, d_oy, oy_pitch, d_u, d_v, c_pitch, d_ou, d_ov, oc_pitch,
                                                    cw, ch, ocw, och, stream), "raisr_hip_process_frame_device")

    def process_host(self, y, oy, u=None, ou=None, v=None, ov=None):
        def pp(a):
            return (a.ctypes.data, a.strides[0]) if a is not None else (None, 0)
        args = [*pp(y), *pp(oy), *pp(u), *pp(ou), *pp(v), *pp(ov)]
        cw, ch = (u.shape[1], u.shape[0]) if u is not None else (0, 0)
        ocw, och = (ou.shape[1], ou.shape[0]) if ou is not None else (0, 0)
        self._check(self._L.raisr_hip_process_host(self._h, *args, cw, ch, ocw, och), "raisr_hip_process_host")

    def set_blending(self, blending):
        self._check(self._L.raisr_hip_set_blending(self._h, blending), "raisr_hip_set_blending")

    def synchronize(self):
        self._check(self._L.raisr_hip_synchronize(self._h), "raisr_hip_synchronize")

    def keep_stages(self, on=True):
        self._check(self._hook("raisr_hip_debug_keep_stages")(self._h, int(on)), "debug_keep_stages")

    def read_stage(self, pass_index=0):
        mode2 = self.cfg.passes == 2 and self.cfg.two_pass_mode == 2
        w = self.cfg.in_width if (pass_index == 0 and mode2) else self.cfg.out_width
        h = self.cfg.in_height if (pass_index == 0 and mode2) else self.cfg.out_height
        hs = np.zeros((h, w), np.uint8); hr = np.zeros((h, w), np.float32)
        self._check(self._hook("raisr_hip_debug_read_stage")(self._h, pass_index, hs.ctypes.data, hr.ctypes.data), "debug_read_stage")
        return hs, hr

    def debug_hash(self, abd, pass_index=0, flavour=HASH_AVX512):
        """Hash buckets of an (n, 3) float32 array of structure-tensor triples, via the device hash functions."""
        abd = np.ascontiguousarray(abd, np.float32)
        out = np.zeros(abd.shape[0], np.uint8)
        self._check(self._hook("raisr_hip_debug_hash")(self._h, pass_index, flavour, abd.ctypes.data, abd.shape[0], out.ctypes.data), "debug_hash")
        return out

    def debug_fold16_check(self, pass_index=0):
        """(disagreements, pairs compared) of the binary16 hash's folded thresholds vs the divisions, exhaustive on the device"""
        out = (ctypes.c_ulonglong * 2)()
        self._check(self._hook("raisr_hip_debug_fold16_check")(self._h, pass_index, out), "debug_fold16_check")
        return int(out[0]), int(out[1])

    def debug_approx_hash(self, abd, pass_index=0, flavour=HASH_AVX512):
        """(bucket, certified, eps) of the certified hash stage for an (n, 3) float32 array of approximate tensor triples."""
        abd = np.ascontiguousarray(abd, np.float32)
        n = abd.shape[0]
        bucket = np.zeros(n, np.uint8); cert = np.zeros(n, np.uint8); eps = ctypes.c_float()
        self._check(self._hook("raisr_hip_debug_approx_hash")(self._h, pass_index, flavour, abd.ctypes.data, n, bucket.ctypes.data, cert.ctypes.data,
                                                 ctypes.byref(eps)), "debug_approx_hash")
        return bucket, cert.astype(bool), float(eps.value)

    def use_streams(self, compute=None, upload=None, download=None):
        """Run the host-plane entry points on caller-owned hipStream_t handles (all three or none; None restores the context's own)."""
        self._check(self._L.raisr_hip_use_streams(self._h, compute, upload, download), "raisr_hip_use_streams")

    def set_fast(self, on=1):
        """NON-bit-exact fast mode. 1: exact buckets, filter stage on the matrix cores (binary16 coefficients);
        2: also keeps the approximate tensor's bucket where the hash stage cannot certify it (no exact re-hash)."""
        self._check(self._L.raisr_hip_set_fast(self._h, int(on)), "set_fast")

    def fast(self):
        return int(self._L.raisr_hip_get_fast(self._h))

    def read_c1tab(self):
        """The class-1 sign table of the certified hash stage as this context built it (65 536 bytes; docs/CERTIFY.md s9)."""
        out = np.zeros(65536, np.uint8)
        self._check(self._hook("raisr_hip_debug_read_c1tab")(self._h, out.ctypes.data), "debug_read_c1tab")
        return out

    def certify_debug(self, collect=True, check=False):
        """Certified hash stage: start (and zero) / stop the statistics; check=True also runs the exact path for every pixel."""
        self._check(self._hook("raisr_hip_debug_certify")(self._h, int(collect), int(check)), "debug_certify")

    def certify_stats(self):
        """dict(uncertain, mismatches, pixels; tiles_listed, tiles_overflow, tiles, tiles_flat) accumulated since certify_debug(True, ...).
        The tile counters describe the worklist (tiles with a non-empty list / whose list overflowed into the all-exact stage): read them
        from a run WITHOUT the self-check, which lists every pixel."""
        out = (ctypes.c_uint * 8)()
        self._check(self._hook("raisr_hip_debug_certify_stats")(self._h, out), "debug_certify_stats")
        return {"uncertain": int(out[0]), "mismatches": int(out[1]), "pixels": int(out[2]),
                "tiles_listed": int(out[3]), "tiles_overflow": int(out[4]), "tiles": int(out[5]), "tiles_flat": int(out[6])}

    def timing_enable(self, on=True):
        self._check(self._L.raisr_hip_kernel_timing_enable(self._h, int(on)), "kernel_timing_enable")

    def timing_read(self, max_kernels=16):
        names = ctypes.create_string_buffer(64 * max_kernels)
        ms = (ctypes.c_float * max_kernels)(); cnt = (ctypes.c_int * max_kernels)()
        n = self._L.raisr_hip_kernel_timing_read(self._h, names, ms, cnt, max_kernels)
        if n < 0:
            raise RuntimeError(last_error())
        out = {}
        for i in range(n):
            nm = names.raw[64 * i:64 * (i + 1)].split(b"\0")[0].decode()
            out[nm] = {"total_ms": float(ms[i]), "count": int(cnt[i])}
        return out


# ----------------------------------------------------------------------------------------------
# Streamed host pipeline (raisr_hip_stream_*): ring of contexts, page-locked planes
# ----------------------------------------------------------------------------------------------
class PinnedPlane:
    """A 2-D numpy array over page-locked host memory from raisr_hip_host_alloc (freed on close / garbage collection)."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib().raisr_hip_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError("raisr_hip_host_alloc failed")
        buf = (ctypes.c_uint8 * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib().raisr_hip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedFrame:
    """Y, U, V planes of one frame in ONE page-locked allocation laid out as raisr_hip_packed_frame_layout says: the library
    moves such a frame with a single PCIe copy."""

    def __init__(self, y_w, y_h, c_w, c_h, bits):
        offs = (ctypes.c_size_t * 3)(); total = ctypes.c_size_t()
        _check(lib().raisr_hip_packed_frame_layout(y_w, y_h, c_w, c_h, bits, offs, ctypes.byref(total)), "packed_frame_layout")
        dt = np.uint8 if bits == 8 else np.uint16
        self._blk = PinnedPlane((total.value,), np.uint8)
        raw = self._blk.array
        isz = np.dtype(dt).itemsize
        self.y = raw[offs[0]:offs[0] + y_w * y_h * isz].view(dt).reshape(y_h, y_w)
        self.u = raw[offs[1]:offs[1] + c_w * c_h * isz].view(dt).reshape(c_h, c_w)
        self.v = raw[offs[2]:offs[2] + c_w * c_h * isz].view(dt).reshape(c_h, c_w)

    def close(self):
        self.y = self.u = self.v = None
        self._blk.close()


class RaisrStream:
    """depth frames in flight per GPU: submit() enqueues, collect() waits for the oldest frame.  `device`: one ordinal, or a list
    of ordinals (raisr_hip_stream_create_multi: frame i runs on device[i % n], one host thread drives them all)."""

    def __init__(self, device, folder, in_w, in_h, out_w, out_h, bits=8, full_range=False, passes=1, mode=1,
                 hash_variant=HASH_AVX512, blending=BLEND_COUNT, chroma=None, depth=4, tie=TIE_HALF_UP, blobs=None):
        """`blobs`: per-pass (device pointer, bytes) of packed model blobs already in device memory (the multi-GPU start-up:
        rank 0 packs, RCCL broadcasts); otherwise the model is read from `folder`."""
        self._h = ctypes.c_void_p()
        if isinstance(device, (list, tuple)):
            arr = (ctypes.c_int * len(device))(*device)
            _check(lib().raisr_hip_stream_create_multi(ctypes.byref(self._h), arr, len(device), depth), "raisr_hip_stream_create_multi")
        else:
            _check(lib().raisr_hip_stream_create(ctypes.byref(self._h), device, depth), "raisr_hip_stream_create")
        self.depth = int(lib().raisr_hip_stream_depth(self._h))      # frames in flight: (at most 4) lanes per device x devices
        try:
            for p in range(passes):
                if blobs is not None:
                    _check(lib().raisr_hip_stream_set_model_blob_device(self._h, p, blobs[p][0], blobs[p][1], None),
                           "raisr_hip_stream_set_model_blob_device")
                    continue
                bank, qstr, qcoh, qa = read_model_folder(folder, bits, p + 1)
                bank = np.ascontiguousarray(bank, np.float32)
                qstr = np.ascontiguousarray(qstr, np.float64); qcoh = np.ascontiguousarray(qcoh, np.float64)
                hk, pt, _ = bank.shape
                _check(lib().raisr_hip_stream_set_model(self._h, p, bank.ctypes.data, hk, pt, qstr.ctypes.data, qcoh.ctypes.data, qa),
                       "raisr_hip_stream_set_model")
            lo, hi = clamp_range(bits, full_range)
            c = RaisrHipConfig()
            c.in_width, c.in_height, c.out_width, c.out_height = in_w, in_h, out_w, out_h
            c.bits, c.clamp_lo, c.clamp_hi = bits, lo, hi
            c.passes, c.two_pass_mode = passes, mode
            c.hash_variant, c.blending = hash_variant, blending
            c.use_pixel_type = int(out_w == 2 * in_w and out_h == 2 * in_h)
            c.tie_rule = tie
            _check(lib().raisr_hip_stream_configure(self._h, ctypes.byref(c)), "raisr_hip_stream_configure")
        except Exception:
            self.close()
            raise
        self.chroma = chroma            # (in_w, in_h, out_w, out_h) of each chroma plane, or None

    def set_fast(self, level):
        _check(lib().raisr_hip_stream_set_fast(self._h, int(level)), "raisr_hip_stream_set_fast")

    def submit(self, y, u, v, oy, ou, ov):
        def pp(a):
            return (a.ctypes.data, a.strides[0]) if a is not None else (None, 0)
        cw, ch, ocw, och = self.chroma if (self.chroma and u is not None) else (0, 0, 0, 0)
        _check(lib().raisr_hip_stream_submit(self._h, *pp(y), *pp(oy), *pp(u), *pp(ou), *pp(v), *pp(ov), cw, ch, ocw, och),
               "raisr_hip_stream_submit")

    def collect(self):
        _check(lib().raisr_hip_stream_collect(self._h), "raisr_hip_stream_collect")

    def in_flight(self):
        return lib().raisr_hip_stream_in_flight(self._h)

    def close(self):
        if self._h:
            lib().raisr_hip_stream_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
