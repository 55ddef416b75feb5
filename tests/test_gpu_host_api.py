"""-m gpu: the reference plugin API (RNLHandler_*) end to end on the GPU -- vf_raisr's call protocol,
chroma planes, row steps larger than the width, repeated frames, re-init -- against the oracle."""
import os

import numpy as np
import pytest

from common import folder, oracle_y

pytestmark = pytest.mark.gpu


def _strided(a, pad):
    buf = np.zeros((a.shape[0], a.shape[1] + pad), a.dtype)
    buf[:, :a.shape[1]] = a
    return buf[:, :a.shape[1]]


@pytest.mark.parametrize("bits,asm,passes,mode", [(8, 2, 1, 1), (8, 5, 2, 2), (10, 2, 1, 1), (8, 6, 2, 1)])
def test_rnlhandler_protocol_yuv420(bits, asm, passes, mode):
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h = 160, 90
    fold = "filters_2x/filters_highres" if not (asm == 5 and mode == 2) else "filters_2x/filters_denoise"
    dt = np.uint8 if bits == 8 else np.uint16
    ys = [synth.natural_y(w, h, bits, seed=s) for s in (1, 2, 3)]
    u = (synth.random_y(w // 2, h // 2, bits, seed=8)).astype(dt)
    v = (synth.random_y(w // 2, h // 2, bits, seed=9)).astype(dt)
    # planes with step > width*bytes, as FFmpeg hands them over (vf_raisr.c:262-285)
    yin = [_strided(y, 24) for y in ys]
    uin, vin = _strided(u, 8), _strided(v, 8)
    oy = _strided(np.zeros((2 * h, 2 * w), dt), 40)
    ou, ov = _strided(np.zeros((h, w), dt), 16), _strided(np.zeros((h, w), dt), 16)
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, bits, R.VideoRange, 20, asm, passes, mode) == 0
    try:
        assert R.RNLHandler_SetRes((yin[0], uin, vin), (oy, ou, ov)) == 0
        ref_asm = 2 if asm == 6 else asm
        for y, ysrc in zip(yin, ys):
            assert R.RNLHandler_Process((y, uin, vin), (oy, ou, ov), R.CountOfBitsChanged) == 0
            ref = oracle_y(ysrc, ("x", fold, (2, 1), bits, passes, mode, ref_asm, False))
            assert np.array_equal(oy, ref)
            assert np.array_equal(ou, O.resize(u, w, h).astype(dt)) and np.array_equal(ov, O.resize(v, w, h).astype(dt))
        assert R.RNLHandler_Process((yin[0], uin, vin), (oy, ou, ov), R.Randomness) == 0
        assert R.RNLHandler_Process((yin[0], uin, vin), (oy, ou, ov), 3) == R.RNLErrorBadParameter
    finally:
        assert R.RNLHandler_Deinit() == 0
    # re-init with a different geometry / ratio in the same process (the reference is a process-global singleton too)
    y = synth.natural_y(96, 64, 8, seed=4)
    c = synth.chroma(48, 32, 8)
    oy2, ou2, ov2 = R.upscale_frame_host(y, c, c, folder("filters_1.5x/filters_highres"), ratio=1.5, bits=8, asm_type=R.AVX512)
    assert np.array_equal(oy2, oracle_y(y, ("x", "filters_1.5x/filters_highres", (3, 2), 8, 1, 1, 2, False)))
    assert oy2.shape == (96, 144) and np.all(ou2 == 128)


def test_model_blob_device_path_matches_file_path():
    """rank-0-reads / others-receive-the-blob path: a context fed through set_model_blob_device gives
    the same output as one fed from the files."""
    import raisr_hip as R
    import synth
    import torch
    y = synth.natural_y(128, 72, 8, seed=6)
    outs = []
    for use_blob in (False, True):
        dev = R.RaisrDevice(0)
        if use_blob:
            for p in range(2):
                bank, qs, qc, qa = R.read_model_folder(folder("filters_2x/filters_highres"), 8, p + 1)
                blob = torch.from_numpy(R.pack_model_blob(bank, qs, qc, qa)).cuda()
                dev.set_model_blob_device(p, blob.data_ptr(), blob.numel())
        else:
            dev.set_model_from_folder(folder("filters_2x/filters_highres"), 8, 2)
        dev.configure(128, 72, 256, 144, bits=8, passes=2, mode=1)
        out = np.zeros((144, 256), np.uint8)
        dev.process_host(y, out)
        dev.close()
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])


def test_loud_failures():
    import raisr_hip as R
    dev = R.RaisrDevice(0)
    with pytest.raises(RuntimeError, match="model not set"):
        dev.configure(64, 64, 128, 128)
    dev.set_model_from_folder(folder("filters_2x/filters_highres"), 8, 1)
    with pytest.raises(RuntimeError, match="pixel types"):
        dev.configure(64, 64, 96, 96, ratio2=False)
    with pytest.raises(RuntimeError, match="binary16"):      # 16-bit samples are not exact in binary16 (8 and 10 bit are supported)
        dev.configure(64, 64, 128, 128, bits=16, full_range=True, hash_variant=R.HASH_FP16)
    dev.close()


def test_device_resident_yuv_frame_matches_host_path():
    import raisr_hip as R
    import synth
    import torch
    w, h = 128, 72
    y = synth.natural_y(w, h, 8, seed=31)
    u = synth.random_y(w // 2, h // 2, 8, seed=32)
    v = synth.random_y(w // 2, h // 2, 8, seed=33)
    dev = R.RaisrDevice(0)
    dev.set_model_from_folder(folder("filters_2x/filters_highres"), 8, 1)
    dev.configure(w, h, 2 * w, 2 * h)
    oy = np.zeros((2 * h, 2 * w), np.uint8); ou = np.zeros((h, w), np.uint8); ov = np.zeros((h, w), np.uint8)
    dev.process_host(y, oy, u, ou, v, ov)
    t = lambda a: torch.from_numpy(a).cuda()
    dy, du, dv = t(y), t(u), t(v)
    doy = torch.zeros((2 * h, 2 * w), dtype=torch.uint8, device="cuda")
    dou, dov = torch.zeros((h, w), dtype=torch.uint8, device="cuda"), torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    dev.process_frame(dy.data_ptr(), w, doy.data_ptr(), 2 * w, du.data_ptr(), dv.data_ptr(), w // 2,
                      dou.data_ptr(), dov.data_ptr(), w, w // 2, h // 2, w, h, s.cuda_stream)
    s.synchronize()
    dev.close()
    assert np.array_equal(doy.cpu().numpy(), oy) and np.array_equal(dou.cpu().numpy(), ou) and np.array_equal(dov.cpu().numpy(), ov)



@pytest.mark.parametrize("layout,cw_div,ch_div", [("yuv422p", 2, 1), ("yuv444p", 1, 1)])
def test_rnlhandler_chroma_layouts(layout, cw_div, ch_div):
    """vf_raisr also accepts 4:2:2 and 4:4:4 (ffmpeg/vf_raisr.c:158-162): the chroma planes simply have other sizes;
    each goes through the cheap upscale at its own size."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h = 96, 64
    y = synth.natural_y(w, h, 8, seed=12)
    u = synth.random_y(w // cw_div, h // ch_div, 8, seed=3)
    v = synth.random_y(w // cw_div, h // ch_div, 8, seed=4)
    oy = np.zeros((2 * h, 2 * w), np.uint8)
    ou = np.zeros((2 * h // ch_div, 2 * w // cw_div), np.uint8); ov = np.zeros_like(ou)
    assert R.RNLHandler_Init(folder("filters_2x/filters_highres"), 2.0, 8, R.VideoRange, 20, R.AVX512, 1, 1) == 0
    try:
        assert R.RNLHandler_SetRes((y, u, v), (oy, ou, ov)) == 0
        assert R.RNLHandler_Process((y, u, v), (oy, ou, ov), R.CountOfBitsChanged) == 0
    finally:
        assert R.RNLHandler_Deinit() == 0
    assert np.array_equal(oy, oracle_y(y, ("x", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False)))
    assert np.array_equal(ou, O.resize(u, ou.shape[1], ou.shape[0]).astype(np.uint8))
    assert np.array_equal(ov, O.resize(v, ov.shape[1], ov.shape[0]).astype(np.uint8))


def test_context_lifecycle_does_not_leak_device_memory():
    """FFmpeg re-creates the filter per stream: Init/SetRes/Process/Deinit cycles and raw context create/destroy must
    hand all HBM back (free memory before == after, within allocator granularity)."""
    import raisr_hip as R
    import synth
    import torch
    torch.cuda.synchronize()
    y = synth.natural_y(320, 180, 8, seed=1)
    c = synth.chroma(160, 90, 8)

    def cycle():
        oy, ou, ov = R.upscale_frame_host(y, c, c, folder("filters_2x/filters_highres"), ratio=2.0, bits=8, asm_type=R.AVX512, passes=2, mode=1)
        dev = R.RaisrDevice(0)
        dev.set_model_from_folder(folder("filters_2x/filters_highres"), 8, 1)
        dev.configure(320, 180, 640, 360, bits=8)
        out = np.zeros((360, 640), np.uint8)
        dev.process_host(y, out)
        dev.close()
        return oy, out

    cycle()                                   # first use pays one-time runtime allocations
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)


def test_rnlprocess_rejects_frames_that_differ_from_setres_geometry():
    """The host copies are asynchronous DMAs sized by RNLSetRes: a smaller plane, or a step shorter than a row, must be
    refused instead of read/written out of bounds (the reference trusts the caller here)."""
    import raisr_hip as R
    import synth
    w, h = 96, 64
    y = synth.natural_y(w, h, 8, seed=1)
    c = synth.chroma(w // 2, h // 2, 8)
    oy = np.zeros((2 * h, 2 * w), np.uint8); ou = np.zeros((h, w), np.uint8); ov = np.zeros((h, w), np.uint8)
    assert R.RNLHandler_Init(folder("filters_2x/filters_highres"), 2.0, 8, R.VideoRange, 20, R.AVX512, 1, 1) == 0
    try:
        assert R.RNLHandler_SetRes((y, c, c), (oy, ou, ov)) == 0
        assert R.RNLHandler_Process((y, c, c), (oy, ou, ov)) == 0
        small = synth.natural_y(w - 16, h, 8, seed=1)
        assert R.RNLHandler_Process((small, c, c), (oy, ou, ov)) == R.RNLErrorBadParameter
        short = synth.natural_y(w, h - 2, 8, seed=1)
        assert R.RNLHandler_Process((short, c, c), (oy, ou, ov)) == R.RNLErrorBadParameter
        assert R.RNLHandler_Process((y, c, c), (oy[:-2], ou, ov)) == R.RNLErrorBadParameter
        assert R.RNLHandler_Process((y, c, c[:, :-2]), (oy, ou, ov)) == R.RNLErrorBadParameter
        # step shorter than one row of samples
        d = [R._vdt(p) for p in (y, c, c, oy, ou, ov)]
        d[0].step = w - 1
        import ctypes
        assert R.lib().RNLHandler_Process(*[ctypes.byref(x) for x in d], R.CountOfBitsChanged) == R.RNLErrorBadParameter
        assert R.RNLHandler_Process((y, c, c), (oy, ou, ov)) == 0          # still usable afterwards
    finally:
        assert R.RNLHandler_Deinit() == 0


def test_failed_reinit_leaves_the_library_uninitialised(tmp_path):
    """A second RNLInit that fails must not leave the first one's device context running with the new parameters."""
    import raisr_hip as R
    import synth
    w, h = 64, 48
    y = synth.natural_y(w, h, 8, seed=2)
    c = synth.chroma(w // 2, h // 2, 8)
    oy = np.zeros((2 * h, 2 * w), np.uint8); ou = np.zeros((h, w), np.uint8); ov = np.zeros((h, w), np.uint8)
    assert R.RNLHandler_Init(folder("filters_2x/filters_highres"), 2.0, 8, R.VideoRange, 20, R.AVX512, 1, 1) == 0
    assert R.RNLHandler_SetRes((y, c, c), (oy, ou, ov)) == 0
    assert R.RNLHandler_Init(str(tmp_path / "no_such_folder"), 2.0, 10, R.FullRange, 20, R.AVX512, 2, 1) != 0
    assert R.RNLHandler_SetRes((y, c, c), (oy, ou, ov)) == R.RNLErrorBadParameter
    assert R.RNLHandler_Process((y, c, c), (oy, ou, ov)) == R.RNLErrorBadParameter
    assert R.RNLHandler_Deinit() == 0


@pytest.mark.parametrize("passes,mode", [(1, 1), (2, 1), (2, 2)])
def test_16bit_depth_against_the_oracle(tmp_path, passes, mode):
    """bits=16 (Raisr.cpp:1462-1469: `_16` files, NF_16 Gaussian table, clamp 0..65535).  No `_16` model ships, so the
    test synthesises the folder from the 10-bit files; u16 full-range frames through oracle and HIP, 1- and 2-pass."""
    import shutil
    import raisr_hip as R
    import synth
    src = folder("filters_2x/filters_highres")
    dst = tmp_path / "filters_16"
    dst.mkdir()
    shutil.copy(f"{src}/config", dst / "config")
    for stem in ("filterbin_2", "Qfactor_strbin_2", "Qfactor_cohbin_2"):
        for sfx in ("", "_2"):
            shutil.copy(f"{src}/{stem}_10{sfx}", dst / f"{stem}_16{sfx}")
    w, h = 88, 56
    rng = np.random.default_rng(16)
    frames = {"random": rng.integers(0, 65536, (h, w)).astype(np.uint16),
              "smooth": np.clip(synth.natural_y(w, h, 10, seed=3).astype(np.int64) * 64 + rng.integers(-300, 300, (h, w)), 0, 65535).astype(np.uint16)}
    for nm, y in frames.items():
        ref = oracle_y(y, ("x", str(dst), (2, 1), 16, passes, mode, 2, True))
        assert ref.dtype == np.uint16 and int(ref.max()) > 1023, nm
        dev = R.RaisrDevice(0)
        try:
            dev.set_model_from_folder(str(dst), 16, passes)
            dev.configure(w, h, 2 * w, 2 * h, bits=16, full_range=True, passes=passes, mode=mode, hash_variant=R.HASH_AVX512)
            out = np.zeros((2 * h, 2 * w), np.uint16)
            dev.process_host(y, out)
        finally:
            dev.close()
        bad = np.argwhere(out != ref)
        assert bad.size == 0, (nm, len(bad), bad[:5].tolist())
    # and through the plugin API (RNLInit's `_16` path names, full-range clamp)
    c = np.full((h // 2, w // 2), 32768, np.uint16)
    oy, _, _ = R.upscale_frame_host(frames["random"], c, c, str(dst), ratio=2.0, bits=16, range_type=R.FullRange, asm_type=R.AVX512,
                                    passes=passes, mode=mode)
    assert np.array_equal(oy, oracle_y(frames["random"], ("x", str(dst), (2, 1), 16, passes, mode, 2, True)))


@pytest.mark.parametrize("bits,passes,use_stream,shift", [(8, 1, False, 0), (10, 2, False, 0), (8, 1, True, 0), (10, 1, False, 6), (10, 2, True, 6)])
def test_hipexternal_device_planes_against_the_oracle(bits, passes, use_stream, shift):
    """asm = HIPExternal (RaisrDefaults.h): SetRes / Process take DEVICE pointers (pitched planes) -- the zero-copy
    counterpart of vf_raisr_opencl.c.  Output compared with the oracle (Y) and the oracle's cheap upscale (chroma).
    shift = 6: P010-style surfaces, every sample stored as value << 6 and VideoDataType::bitShift = 6 on all six planes (what
    vf_raisr_opencl.c:111,117 passes): the library shifts down on the way in and up on the way out, as the reference's OpenCL
    pre/post-process kernels do."""
    import ctypes
    import oracle_py as O
    import raisr_hip as R
    import synth
    import torch
    w, h = 176, 100
    fold = "filters_2x/filters_highres"
    tdt = torch.uint8 if bits == 8 else torch.uint16
    ndt = np.uint8 if bits == 8 else np.uint16
    bps = 1 if bits == 8 else 2
    y = synth.natural_y(w, h, bits, seed=21)
    u = synth.random_y(w // 2, h // 2, bits, seed=22).astype(ndt)
    v = synth.random_y(w // 2, h // 2, bits, seed=23).astype(ndt)

    def dev_plane(a, pad):                                  # pitched device plane holding `a` (MSB-aligned by `shift` bits)
        host = np.zeros((a.shape[0], a.shape[1] + pad), ndt)
        host[:, :a.shape[1]] = a.astype(ndt) << shift
        return torch.from_numpy(host.view(np.int16) if bits != 8 else host).cuda().view(tdt)

    def vdt(t, width, height):
        d = R.VideoDataType()
        d.pData = t.data_ptr(); d.width = width; d.height = height; d.step = t.stride(0) * bps; d.bitShift = shift
        return d
    dy, du, dv = dev_plane(y, 16), dev_plane(u, 8), dev_plane(v, 8)
    oy = torch.zeros((2 * h, 2 * w + 32), dtype=tdt, device="cuda")
    ou = torch.zeros((h, w + 16), dtype=tdt, device="cuda")
    ov = torch.zeros((h, w + 16), dtype=tdt, device="cuda")
    ds = [vdt(dy, w, h), vdt(du, w // 2, h // 2), vdt(dv, w // 2, h // 2), vdt(oy, 2 * w, 2 * h), vdt(ou, w, h), vdt(ov, w, h)]
    if use_stream:
        # the reference reads bitShift from the INPUT descriptors only (Raisr.cpp:1313-1348): a caller that leaves the output
        # descriptors' field at 0 is served with the input's alignment
        for d in ds[3:]:
            d.bitShift = 0
    refs = [ctypes.byref(x) for x in ds]
    stream = torch.cuda.Stream() if use_stream else None
    assert R.RNLHandler_SetOpenCLContext(0, 0, stream.cuda_stream if stream else None) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, bits, R.VideoRange, 20, R.HIPExternal, passes, 1) == 0
    try:
        torch.cuda.synchronize()
        assert R.lib().RNLHandler_SetRes(*refs) == 0
        for _ in range(2):
            assert R.lib().RNLHandler_Process(*refs, R.CountOfBitsChanged) == 0
        if stream:
            stream.synchronize()                           # stream-ordered: the caller's stream carries the work
    finally:
        assert R.RNLHandler_Deinit() == 0
        R.RNLHandler_SetOpenCLContext(0, 0, None)
    got_y = oy[:, :2 * w].cpu().numpy().view(ndt)
    ref = oracle_y(y, ("x", fold, (2, 1), bits, passes, 1, 2, False))
    assert np.array_equal(got_y, ref.astype(ndt) << shift)
    assert np.array_equal(ou[:, :w].cpu().numpy().view(ndt), O.resize(u, w, h).astype(ndt) << shift)
    assert np.array_equal(ov[:, :w].cpu().numpy().view(ndt), O.resize(v, w, h).astype(ndt) << shift)


@pytest.mark.parametrize("bits,asm,passes,mode,depth", [(8, 2, 1, 1, 4), (10, 2, 2, 1, 2), (8, 5, 2, 2, 3), (8, 6, 1, 1, 1)])
def test_async_submit_collect_frames_equal_the_oracle_in_order(bits, asm, passes, mode, depth):
    """RNLHandler_Submit / _Collect (the call sequence of the FFmpeg filter with async=N: submit frame n, collect frame n - N,
    drain at EOF): every collected frame is the oracle's frame, in submission order; a full ring refuses the next Submit."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h = 176, 100
    fold = "filters_2x/filters_denoise" if mode == 2 else "filters_2x/filters_highres"
    dt = np.uint8 if bits == 8 else np.uint16
    n = 11
    ys = [synth.natural_y(w, h, bits, seed=100 + s) for s in range(n)]
    us = [synth.random_y(w // 2, h // 2, bits, seed=200 + s).astype(dt) for s in range(n)]
    vs = [synth.random_y(w // 2, h // 2, bits, seed=300 + s).astype(dt) for s in range(n)]
    ref_asm = 2 if asm == 6 else asm
    refs = [oracle_y(y, ("x", fold, (2, 1), bits, passes, mode, ref_asm, False)) for y in ys]
    outs = [(_strided(np.zeros((2 * h, 2 * w), dt), 40), np.zeros((h, w), dt), np.zeros((h, w), dt)) for _ in range(n)]
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, bits, R.VideoRange, 20, asm, passes, mode) == 0
    try:
        assert R.RNLHandler_Submit((ys[0], us[0], vs[0]), outs[0]) == R.RNLErrorBadParameter          # before SetRes
        assert R.RNLHandler_SetRes((ys[0], us[0], vs[0]), outs[0]) == 0
        assert R.RNLHandler_Submit((ys[0], us[0], vs[0]), outs[0]) == R.RNLErrorBadParameter          # no ring asked for
        assert R.RNLHandler_Collect() == R.RNLErrorBadParameter
        assert R.RNLHandler_SetAsyncDepth(5) == R.RNLErrorBadParameter                                # more than the ring builds: refused, not clamped
        assert R.RNLHandler_SetAsyncDepth(depth) == 0
        built = min(depth, 4)
        collected = 0

        def collect():
            nonlocal collected
            assert R.RNLHandler_Collect() == 0
            oy, ou, ov = outs[collected]
            assert np.array_equal(oy, refs[collected]), (collected, int((oy != refs[collected]).sum()))
            assert np.array_equal(ou, O.resize(us[collected], w, h).astype(dt)) and np.array_equal(ov, O.resize(vs[collected], w, h).astype(dt))
            collected += 1
        for i in range(n):
            if R.RNLHandler_FramesInFlight() == built:
                assert R.RNLHandler_Submit((ys[i], us[i], vs[i]), outs[i]) == R.RNLErrorInsufficientResources
                collect()
            assert R.RNLHandler_Submit((ys[i], us[i], vs[i]), outs[i]) == 0
        assert R.RNLHandler_SetAsyncDepth(0) == R.RNLErrorBadParameter                                # frames in flight
        assert R.RNLHandler_SetRes((ys[0], us[0], vs[0]), outs[0]) == R.RNLErrorBadParameter          # ... are not dropped by a SetRes either
        while R.RNLHandler_FramesInFlight():
            collect()
        assert collected == n and R.RNLHandler_Collect() == R.RNLErrorBadParameter
        # the synchronous entry still works next to the ring, and a wrong geometry is refused
        assert R.RNLHandler_Process((ys[1], us[1], vs[1]), outs[0]) == 0 and np.array_equal(outs[0][0], refs[1])
        assert R.RNLHandler_Submit((ys[0][:-2], us[0], vs[0]), outs[0]) == R.RNLErrorBadParameter
        assert R.RNLHandler_SetAsyncDepth(0) == 0
    finally:
        assert R.RNLHandler_Deinit() == 0


def test_opt_in_pin_cache_with_recycled_and_many_live_buffers(monkeypatch):
    """RAISR_HIP_PIN=1 (opt-in; the host keeps every plane buffer allocated until RNLHandler_Deinit): RNLProcess page-locks the
    caller's planes on first sight and remembers them (csrc/raisr_api.cpp: PinCache).  Frames through a recycled pool of
    malloc'ed buffers against the oracle; then more live buffers than the cache holds (48): the least recently used
    registrations are dropped and nothing breaks.  Buffers are freed only after Deinit."""
    import ctypes
    import raisr_hip as R
    import synth
    monkeypatch.setenv("RAISR_HIP_PIN", "1")
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    w, h = 480, 272                                      # planes above the cache's 64 KB threshold
    fold = "filters_2x/filters_highres"
    case = ("x", fold, (2, 1), 8, 1, 1, 2, False)
    ys = [synth.natural_y(w, h, 8, seed=500 + s) for s in range(4)]
    refs = [oracle_y(y, case) for y in ys]
    c = synth.chroma(w // 2, h // 2, 8)
    live = []

    def plane(shape):
        n = shape[0] * shape[1]
        p = libc.malloc(n)
        live.append(p)
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p)).reshape(shape)

    def frame_set():
        return (plane((h, w)), plane((h // 2, w // 2))), (plane((2 * h, 2 * w)), plane((h, w)), plane((h, w)))
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, 8, R.VideoRange, 20, R.AVX512, 1, 1) == 0
    try:
        pool = [frame_set() for _ in range(3)]
        for (ay, au), _ in pool:
            au[...] = c
        assert R.RNLHandler_SetRes((pool[0][0][0], pool[0][0][1], pool[0][0][1]), pool[0][1]) == 0
        for it in range(24):
            (ay, au), (ao, aou, aov) = pool[it % 3]
            ay[...] = ys[it % 4]; ao[...] = 0
            assert R.lib().raisr_hip_host_is_page_locked(ay.ctypes.data) == (1 if it >= 3 else 0)
            assert R.RNLHandler_Process((ay, au, au), (ao, aou, aov)) == 0
            assert np.array_equal(ao, refs[it % 4]), (it, int((ao != refs[it % 4]).sum()))
        for it in range(20):                             # 20 x 5 planes above the threshold: more than the cache has entries
            (ay, au), (ao, aou, aov) = frame_set()
            ay[...] = ys[it % 4]; au[...] = c
            assert R.RNLHandler_Process((ay, au, au), (ao, aou, aov)) == 0
            assert np.array_equal(ao, refs[it % 4]), ("many buffers", it)
    finally:
        assert R.RNLHandler_Deinit() == 0
        for p in live:
            libc.free(p)


@pytest.mark.parametrize("bits,pad", [(8, 0), (8, 48), (10, 16)])
def test_planes_from_hostalloc_take_the_page_locked_path(bits, pad):
    """Frame buffers from RNLHandler_HostAlloc (what ffmpeg/vf_raisr_hip.diff's buffer pools hand to the filter): recognised as
    page-locked, processed with the last pass in row ranges (no RAISR_HIP_* set), same bits for Y and chroma, with and without
    row padding; ordinary numpy planes next to them in the same session; freed after Deinit."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h = 322, 190
    dt = np.uint8 if bits == 8 else np.uint16
    fold = "filters_2x/filters_highres"
    ys = [synth.natural_y(w, h, bits, seed=800 + i) for i in range(3)]
    u = synth.random_y(w // 2, h // 2, bits, seed=7).astype(dt); v = synth.random_y(w // 2, h // 2, bits, seed=8).astype(dt)
    cw, ch = w // 2, h // 2
    keep = [R.HostPlane((h, w), dt, w + pad), R.HostPlane((ch, cw), dt, cw + pad), R.HostPlane((ch, cw), dt, cw + pad),
            R.HostPlane((2 * h, 2 * w), dt, 2 * w + pad), R.HostPlane((2 * ch, 2 * cw), dt, 2 * cw + pad), R.HostPlane((2 * ch, 2 * cw), dt, 2 * cw + pad)]
    iy, iu, iv, oy, ou, ov = [k.array for k in keep]
    assert R.lib().raisr_hip_host_is_page_locked(iy.ctypes.data) == 1 and R.lib().raisr_hip_host_is_page_locked(ys[0].ctypes.data) == 0
    iu[...] = u; iv[...] = v
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, bits, R.VideoRange, 20, R.AVX512, 1, 1) == 0
    try:
        assert R.RNLHandler_SetRes((iy, iu, iv), (oy, ou, ov)) == 0
        for y in ys:
            iy[...] = y; oy[...] = 0; ou[...] = 0
            assert R.RNLHandler_Process((iy, iu, iv), (oy, ou, ov)) == 0
            ref = oracle_y(y, ("x", fold, (2, 1), bits, 1, 1, 2, False))
            assert np.array_equal(oy, ref)
            assert np.array_equal(ou, O.resize(u, 2 * cw, 2 * ch).astype(dt)) and np.array_equal(ov, O.resize(v, 2 * cw, 2 * ch).astype(dt))
            # pageable planes in the same session (mixed: page-locked output, pageable input, and the other way round)
            oy[...] = 0
            assert R.RNLHandler_Process((y, u, v), (oy, ou, ov)) == 0 and np.array_equal(oy, ref)
            noy = np.zeros((2 * h, 2 * w), dt); nou = np.zeros((2 * ch, 2 * cw), dt); nov = np.zeros_like(nou)
            assert R.RNLHandler_Process((iy, iu, iv), (noy, nou, nov)) == 0 and np.array_equal(noy, ref)
    finally:
        assert R.RNLHandler_Deinit() == 0
        for k in keep:
            k.close()                                    # after Deinit: allowed


@pytest.mark.parametrize("chunks", [1, 2, 4])
def test_plugin_path_with_row_chunks_and_strided_planes(chunks, monkeypatch):
    """RNLHandler_Process with the last pass cut into row ranges (RAISR_HIP_CHUNKS): rows are downloaded while later rows are still
    being computed -- same bits, Y and chroma, with steps larger than the rows and a frame height that is no multiple of anything."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    monkeypatch.setenv("RAISR_HIP_CHUNKS", str(chunks))
    w, h = 322, 181
    ys = [synth.natural_y(w, h, 8, seed=70 + i) for i in range(3)]
    u = synth.random_y(w // 2, h // 2, 8, seed=5); v = synth.random_y(w // 2, h // 2, 8, seed=6)
    cw, ch = w // 2, h // 2
    oy = _strided(np.zeros((2 * h, 2 * w), np.uint8), 24)
    ou, ov = _strided(np.zeros((2 * ch, 2 * cw), np.uint8), 8), _strided(np.zeros((2 * ch, 2 * cw), np.uint8), 8)
    for passes, fold in ((1, "filters_2x/filters_highres"), (2, "filters_2x/filters_denoise")):
        assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
        assert R.RNLHandler_Init(folder(fold), 2.0, 8, R.VideoRange, 20, R.HIP, passes, 1) == 0
        try:
            assert R.RNLHandler_SetRes((ys[0], u, v), (oy, ou, ov)) == 0
            for y in ys:
                oy[...] = 0
                yin = _strided(y, 10)                     # a fresh buffer per call, freed right after it: hosts do that
                if os.environ.get("RAISR_TEST_TRACE_PTRS"):
                    os.write(1, f"[ptrs] yin={yin.ctypes.data:#x}+{yin.base.nbytes} oy={oy.ctypes.data:#x}+{oy.base.nbytes} ou={ou.ctypes.data:#x} ov={ov.ctypes.data:#x} u={u.ctypes.data:#x}\n".encode())
                assert R.RNLHandler_Process((yin, u, v), (oy, ou, ov)) == 0
                del yin
                assert np.array_equal(oy, oracle_y(y, ("x", fold, (2, 1), 8, passes, 1, 2, False)))
                assert np.array_equal(ou, O.resize(u, 2 * cw, 2 * ch).astype(np.uint8)) and np.array_equal(ov, O.resize(v, 2 * cw, 2 * ch).astype(np.uint8))
            assert R.RNLHandler_Process((ys[0], u, v), (oy, ou, ov), R.Randomness) == 0           # not chunked: same entry, other blend
        finally:
            assert R.RNLHandler_Deinit() == 0


def test_a_fresh_set_of_pageable_buffers_per_call_and_per_submit():
    """What an ordinary host does: allocate the frame's planes, call, free them -- every call, contiguous and strided, through the
    synchronous entry and through Submit / Collect (planes freed right after their Collect).  The bits stay the oracle's.
    (The pattern that preceded the sporadic GPU page fault of rounds 1-2's host path, csrc/host_copy.h; this test did not reproduce
    it with RAISR_HIP_BOUNCE=0 either -- it is here as the regression test of the ordinary-host behaviour.)"""
    import gc
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h = 322, 181
    fold = "filters_2x/filters_highres"
    cw, ch = w // 2, h // 2
    ys = [synth.natural_y(w, h, 8, seed=900 + i) for i in range(4)]
    refs = [oracle_y(y, ("x", fold, (2, 1), 8, 1, 1, 2, False)) for y in ys]
    u0 = synth.random_y(cw, ch, 8, seed=15); v0 = synth.random_y(cw, ch, 8, seed=16)
    ru, rv = O.resize(u0, 2 * cw, 2 * ch).astype(np.uint8), O.resize(v0, 2 * cw, 2 * ch).astype(np.uint8)

    def fresh(it):
        pad = 10 if (it // 15) % 2 == 0 else 0           # the same sizes call after call: the allocator hands the same addresses out again
        mk = (lambda a: _strided(a, pad)) if pad else (lambda a: a.copy())
        return (mk(ys[it % 4]), mk(u0), mk(v0)), (mk(np.zeros((2 * h, 2 * w), np.uint8)), mk(np.zeros((2 * ch, 2 * cw), np.uint8)), mk(np.zeros((2 * ch, 2 * cw), np.uint8)))
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, 8, R.VideoRange, 20, R.HIP, 1, 1) == 0
    try:
        ins, outs = fresh(0)
        assert R.RNLHandler_SetRes(ins, outs) == 0
        for it in range(60):
            ins, outs = fresh(it)
            assert R.RNLHandler_Process(ins, outs) == 0
            assert np.array_equal(outs[0], refs[it % 4]) and np.array_equal(outs[1], ru) and np.array_equal(outs[2], rv), it
            del ins, outs
            if it % 7 == 0:
                gc.collect()
        assert R.RNLHandler_SetAsyncDepth(3) == 0
        queue = []
        for it in range(45):
            if len(queue) == 3:
                j, ins, outs = queue.pop(0)
                assert R.RNLHandler_Collect() == 0
                assert np.array_equal(outs[0], refs[j % 4]) and np.array_equal(outs[1], ru) and np.array_equal(outs[2], rv), j
                del ins, outs
            ins, outs = fresh(it)
            assert R.RNLHandler_Submit(ins, outs) == 0
            queue.append((it, ins, outs))
        while queue:
            j, ins, outs = queue.pop(0)
            assert R.RNLHandler_Collect() == 0
            assert np.array_equal(outs[0], refs[j % 4]), j
    finally:
        assert R.RNLHandler_Deinit() == 0


@pytest.mark.parametrize("bits,ratio", [(8, (2, 1)), (10, (2, 1)), (8, (3, 2))])
def test_hipexternal_nv12_style_interleaved_chroma(bits, ratio):
    """Device frames as hardware decoders produce them (NV12 / P010: one chroma plane of interleaved (U, V) pairs) -- what
    ffmpeg/vf_raisr_hipframes.c hands over: both chroma descriptors carry RAISR_HIP_INTERLEAVED2 and point one sample apart into
    the same plane.  Y against the oracle, each chroma channel against the oracle's cheap upscale; padding bytes untouched."""
    import ctypes
    import oracle_py as O
    import raisr_hip as R
    import synth
    import torch
    rn, rd = ratio
    w, h = 176, 100
    ow, oh = w * rn // rd, h * rn // rd
    cw, ch, ocw, och = w // 2, h // 2, ow // 2, oh // 2
    fold = "filters_2x/filters_highres" if rn == 2 else "filters_1.5x/filters_highres"
    tdt = torch.uint8 if bits == 8 else torch.uint16
    ndt = np.uint8 if bits == 8 else np.uint16
    bps = 1 if bits == 8 else 2
    FLAG = 0x80000000
    y = synth.natural_y(w, h, bits, seed=41)
    u = synth.random_y(cw, ch, bits, seed=42).astype(ndt)
    v = synth.random_y(cw, ch, bits, seed=43).astype(ndt)
    uv = np.zeros((ch, 2 * cw + 12), ndt)
    uv[:, 0:2 * cw:2] = u; uv[:, 1:2 * cw:2] = v

    def to_dev(a):
        return torch.from_numpy(a.view(np.int16) if bits != 8 else a).cuda().view(tdt)
    dy = to_dev(np.ascontiguousarray(y)); duv = to_dev(uv)
    oy = torch.zeros((oh, ow + 8), dtype=tdt, device="cuda")
    ouv = torch.full((och, 2 * ocw + 20), 7, dtype=tdt, device="cuda")

    def vdt(ptr, width, height, step, shift=0):
        d = R.VideoDataType()
        d.pData = ptr; d.width = width; d.height = height; d.step = step; d.bitShift = shift
        return d
    ds = [vdt(dy.data_ptr(), w, h, dy.stride(0) * bps),
          vdt(duv.data_ptr(), cw, ch, duv.stride(0) * bps, FLAG), vdt(duv.data_ptr() + bps, cw, ch, duv.stride(0) * bps, FLAG),
          vdt(oy.data_ptr(), ow, oh, oy.stride(0) * bps),
          vdt(ouv.data_ptr(), ocw, och, ouv.stride(0) * bps, FLAG), vdt(ouv.data_ptr() + bps, ocw, och, ouv.stride(0) * bps, FLAG)]
    refs = [ctypes.byref(x) for x in ds]
    assert R.RNLHandler_SetOpenCLContext(0, 0, None) == 0
    assert R.RNLHandler_Init(folder(fold), rn / rd, bits, R.VideoRange, 20, R.HIPExternal, 1, 1) == 0
    try:
        torch.cuda.synchronize()
        assert R.lib().RNLHandler_SetRes(*refs) == 0
        assert R.lib().RNLHandler_Process(*refs, R.CountOfBitsChanged) == 0
        ds[2].bitShift = 0                                   # the flag on one side only: refused
        assert R.lib().RNLHandler_Process(*refs, R.CountOfBitsChanged) == R.RNLErrorBadParameter
    finally:
        assert R.RNLHandler_Deinit() == 0
    torch.cuda.synchronize()
    assert np.array_equal(oy[:, :ow].cpu().numpy().view(ndt), oracle_y(y, ("x", fold, ratio, bits, 1, 1, 2, False)))
    got = ouv.cpu().numpy().view(ndt)
    assert np.array_equal(got[:, 0:2 * ocw:2], O.resize(u, ocw, och).astype(ndt))
    assert np.array_equal(got[:, 1:2 * ocw:2], O.resize(v, ocw, och).astype(ndt))
    assert np.all(got[:, 2 * ocw:] == 7)
