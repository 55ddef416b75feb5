"""-m gpu: the multi-device stream ring (raisr_hip_stream_create_multi) and its plugin-API face (RNLHandler_SetDeviceList /
RAISR_HIP_DEVICES) -- one host thread, frame i on device slot i % n, in-order collection.  The GPU box has ONE device, so the
device list names it several times: every slot still has its own lanes, streams, scratch planes and bounce memory, the model
still reaches the slots through raisr_hip_broadcast_model_blob_devices (copy fall-back: RCCL wants distinct devices), and every
frame must equal the oracle's, in order.  (north_star: "host code stays C++ ... frames shard across the 8 GPUs".)"""
import ctypes

import numpy as np
import pytest

from common import folder, oracle_y

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices,depth", [([0], 2), ([0, 0], 2), ([0, 0, 0], 1), ([0, 0, 0, 0], 4)])
@pytest.mark.parametrize("pinned", [True, False])
def test_600_frame_c5_shaped_stream_over_a_device_list(devices, depth, pinned):
    """C5's shape (2x, highres, 10-bit, yuv420p) at 480 x 270 -> 960 x 540, 600 frames cycling over 5 distinct inputs."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h, bits, n, uniq = 480, 270, 10, 600, 5
    fold = "filters_2x/filters_highres"
    case = ("x", fold, (2, 1), bits, 1, 1, 2, False)
    ys = [synth.natural_y(w, h, bits, seed=500 + i) for i in range(uniq)]
    refs = [oracle_y(y, case) for y in ys]
    cin = synth.random_y(w // 2, h // 2, bits, seed=9).astype(np.uint16)
    cref = O.resize(cin, w, h).astype(np.uint16)
    pins = []

    def plane(shape, fill=None):
        if pinned:
            pl = R.PinnedPlane(shape, np.uint16); pins.append(pl); a = pl.array
        else:
            a = np.zeros(shape, np.uint16)
        if fill is not None:
            a[...] = fill
        return a
    st = R.RaisrStream(devices, folder(fold), w, h, 2 * w, 2 * h, bits=bits, chroma=(w // 2, h // 2, w, h), depth=depth)
    cap = st.depth
    assert cap == len(devices) * depth
    assert R.lib().raisr_hip_stream_device_count(st._h) == len(devices)
    assert [R.lib().raisr_hip_stream_device_of_frame(st._h, i) for i in range(7)] == [devices[i % len(devices)] for i in range(7)]
    yin = [plane((h, w), y) for y in ys]
    c = plane((h // 2, w // 2), cin)
    outs = [(plane((2 * h, 2 * w)), plane((h, w)), plane((h, w))) for _ in range(cap)]
    bad = []
    try:
        done = inflight = 0

        def collect():
            nonlocal done, inflight
            st.collect()
            oy, ou, ov = outs[done % cap]
            if not np.array_equal(oy, refs[done % uniq]) or not np.array_equal(ou, cref) or not np.array_equal(ov, cref):
                bad.append(done)
            oy[0, :8] = 0; ou[0, :8] = 0                       # stale bytes must not survive into the lane's next frame
            done += 1; inflight -= 1
        for i in range(n):
            if inflight == cap:
                with pytest.raises(RuntimeError):
                    st.submit(yin[0], c, c, *outs[0])          # ring full: collect first
                collect()
            st.submit(yin[i % uniq], c, c, *outs[i % cap])
            inflight += 1
        while inflight:
            collect()
    finally:
        st.close()
        for pl in pins:
            pl.close()
    assert done == n and not bad, bad[:10]


def test_create_multi_argument_checks():
    import raisr_hip as R
    h = ctypes.c_void_p()
    L = R.lib()
    one = (ctypes.c_int * 1)(0)
    assert L.raisr_hip_stream_create_multi(ctypes.byref(h), one, 0, 2) != 0
    assert L.raisr_hip_stream_create_multi(ctypes.byref(h), one, 1, 5) != 0                     # lanes per device: 1..4
    assert L.raisr_hip_stream_create_multi(ctypes.byref(h), None, 1, 2) != 0
    many = (ctypes.c_int * 17)(*([0] * 17))
    assert L.raisr_hip_stream_create_multi(ctypes.byref(h), many, 17, 1) != 0
    gone = (ctypes.c_int * 2)(0, 99)
    assert L.raisr_hip_stream_create_multi(ctypes.byref(h), gone, 2, 1) != 0                    # no such device: nothing leaks, nothing half-built
    assert not h.value


@pytest.mark.parametrize("how", ["call", "env"])
def test_plugin_api_ring_over_a_device_list(how, monkeypatch):
    """RNLHandler_SetDeviceList("0,0") / RAISR_HIP_DEVICES=0,0 + SetAsyncDepth(2): four frames in flight over two device slots, the
    FFmpeg filter's call sequence with async=2, every collected frame the oracle's, in order."""
    import raisr_hip as R
    import synth
    w, h, bits, n = 176, 100, 8, 23
    fold = "filters_2x/filters_highres"
    ys = [synth.natural_y(w, h, bits, seed=700 + s) for s in range(n)]
    us = [synth.random_y(w // 2, h // 2, bits, seed=800 + s).astype(np.uint8) for s in range(n)]
    refs = [oracle_y(y, ("x", fold, (2, 1), bits, 1, 1, 2, False)) for y in ys]
    outs = [(np.zeros((2 * h, 2 * w), np.uint8), np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)) for _ in range(n)]
    if how == "env":
        monkeypatch.setenv("RAISR_HIP_DEVICES", "0,0")
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder(fold), 2.0, bits, R.VideoRange, 20, R.AVX512, 1, 1) == 0
    try:
        assert R.RNLHandler_SetRes((ys[0], us[0], us[0]), outs[0]) == 0
        assert R.RNLHandler_SetDeviceList("0,7") == R.RNLErrorBadParameter                      # no device 7 on this box
        assert R.RNLHandler_SetDeviceList("0;0") == R.RNLErrorBadParameter
        if how == "call":
            assert R.RNLHandler_SetDeviceList("0,0") == 0
        assert R.RNLHandler_SetAsyncDepth(2) == 0
        assert R.RNLHandler_AsyncCapacity() == 4                                                  # before the ring exists: depth x listed GPUs
        inflight = done = 0
        for i in range(n):
            if inflight == 4:                                                                     # 2 device slots x depth 2
                assert R.RNLHandler_Submit((ys[i], us[i], us[i]), outs[i]) == R.RNLErrorInsufficientResources
                assert R.RNLHandler_Collect() == 0
                assert np.array_equal(outs[done][0], refs[done]), done
                done += 1; inflight -= 1
            assert R.RNLHandler_Submit((ys[i], us[i], us[i]), outs[i]) == 0
            inflight += 1
            if i == 5:
                assert R.RNLHandler_SetDeviceList("0") == R.RNLErrorBadParameter                  # frames in flight: collect first
        while inflight:
            assert R.RNLHandler_Collect() == 0
            assert np.array_equal(outs[done][0], refs[done]), done
            done += 1; inflight -= 1
        assert done == n
        assert R.RNLHandler_SetDeviceList("") == 0                                                # an explicit empty list = the single device again, whatever
        cap = 2                                                                                   # RAISR_HIP_DEVICES says (include/raisr/RaisrHandler.h)
        assert R.RNLHandler_AsyncCapacity() == cap
        for k in range(cap):
            assert R.RNLHandler_Submit((ys[k], us[k], us[k]), outs[k]) == 0
        assert R.RNLHandler_Submit((ys[cap], us[cap], us[cap]), outs[cap]) == R.RNLErrorInsufficientResources
        for k in range(cap):
            assert R.RNLHandler_Collect() == 0
            assert np.array_equal(outs[k][0], refs[k]), k
    finally:
        assert R.RNLHandler_Deinit() == 0


def test_malformed_device_environment_is_an_error_at_set_async_depth(monkeypatch):
    """RAISR_HIP_DEVICES that is not a device list: RNLHandler_SetAsyncDepth(n > 0) refuses (with the library's message) instead of
    reporting a capacity of 0 that a caller would divide its queue by; an explicit RNLHandler_SetDeviceList overrides the variable."""
    import raisr_hip as R
    import synth
    w, h = 176, 100
    y = synth.natural_y(w, h, 8, seed=1); u = synth.chroma(w // 2, h // 2, 8)
    out = (np.zeros((2 * h, 2 * w), np.uint8), np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8))
    monkeypatch.setenv("RAISR_HIP_DEVICES", "0;x")
    assert R.RNLHandler_SetOpenCLContext(0, 0) == 0
    assert R.RNLHandler_Init(folder("filters_2x/filters_highres"), 2.0, 8, R.VideoRange, 20, R.AVX512, 1, 1) == 0
    try:
        assert R.RNLHandler_SetRes((y, u, u), out) == 0
        assert R.RNLHandler_SetAsyncDepth(2) == R.RNLErrorBadParameter
        assert R.RNLHandler_AsyncCapacity() == 0
        assert R.RNLHandler_SetAsyncDepth(0) == 0                                                 # switching the ring off never consults the list
        assert R.RNLHandler_SetDeviceList("0") == 0
        assert R.RNLHandler_SetAsyncDepth(2) == 0 and R.RNLHandler_AsyncCapacity() == 2
        assert R.RNLHandler_Submit((y, u, u), out) == 0 and R.RNLHandler_Collect() == 0
    finally:
        assert R.RNLHandler_Deinit() == 0


def test_blob_broadcast_over_a_device_list_copies_the_blob():
    """raisr_hip_broadcast_model_blob_devices with a repeated device (the copy fall-back) and with RAISR_HIP_NO_RCCL semantics:
    every destination ends up with the source's bytes."""
    import raisr_hip as R
    import torch
    bank, qstr, qcoh, qa = R.read_model_folder(folder("filters_2x/filters_highres"), 8, 1)
    blob = R.pack_model_blob(bank, qstr, qcoh, qa)
    src = torch.from_numpy(blob).cuda()
    dsts = [torch.zeros_like(src) for _ in range(3)]
    devs = (ctypes.c_int * 4)(0, 0, 0, 0)
    ptrs = (ctypes.c_void_p * 4)(src.data_ptr(), *[d.data_ptr() for d in dsts])
    assert R.lib().raisr_hip_broadcast_model_blob_devices(devs, 4, ptrs, blob.size) == 0
    torch.cuda.synchronize()
    for d in dsts:
        assert torch.equal(d, src)
    assert R.lib().raisr_hip_broadcast_model_blob_devices(devs, 0, ptrs, blob.size) != 0


def test_in_process_rccl_broadcast_with_one_rank(monkeypatch):
    """The opt-in RCCL leg of raisr_hip_broadcast_model_blob_devices (ncclCommInitAll once per device list, grouped ncclBroadcast through
    the dlopen'ed librccl) needs distinct devices; a one-GPU box can run it with ONE rank only (RAISR_HIP_FORCE_RCCL=1): the symbols
    resolve, the communicator comes up, a second broadcast (the model's second pass) REUSES it, the root's blob is untouched."""
    import raisr_hip as R
    import torch
    monkeypatch.setenv("RAISR_HIP_FORCE_RCCL", "1")
    bank, qstr, qcoh, qa = R.read_model_folder(folder("filters_2x/filters_highres"), 8, 1)
    blob = R.pack_model_blob(bank, qstr, qcoh, qa)
    src = torch.from_numpy(blob).cuda()
    keep = src.clone()
    devs = (ctypes.c_int * 1)(0)
    ptrs = (ctypes.c_void_p * 1)(src.data_ptr())
    for _ in range(3):
        assert R.lib().raisr_hip_broadcast_model_blob_devices(devs, 1, ptrs, blob.size) == 0, R.last_error()
    torch.cuda.synchronize()
    assert torch.equal(src, keep)


@pytest.mark.parametrize("rccl", [False, True], ids=["peer-copies", "in-process-rccl"])
def test_blob_broadcast_over_distinct_devices(monkeypatch, rccl):
    """Two or more GPUs (skipped on the one-GPU boxes of rounds 1-6): the default hand-over -- concurrent peer copies out of devices[0] --
    and the opt-in in-process RCCL broadcast both leave every device with the source's bytes, twice in a row (pass 1, pass 2)."""
    import raisr_hip as R
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    if rccl:
        monkeypatch.setenv("RAISR_HIP_RCCL", "1")
    bank, qstr, qcoh, qa = R.read_model_folder(folder("filters_2x/filters_highres"), 8, 1)
    blob = R.pack_model_blob(bank, qstr, qcoh, qa)
    src = torch.from_numpy(blob).to("cuda:0")
    dsts = [torch.zeros(blob.size, dtype=torch.uint8, device=f"cuda:{i}") for i in range(1, n)]
    devs = (ctypes.c_int * n)(*range(n))
    ptrs = (ctypes.c_void_p * n)(src.data_ptr(), *[d.data_ptr() for d in dsts])
    for rep in range(2):
        for d in dsts:
            d.zero_()
        torch.cuda.synchronize()
        assert R.lib().raisr_hip_broadcast_model_blob_devices(devs, n, ptrs, blob.size) == 0, R.last_error()
        for d in dsts:
            assert torch.equal(d.cpu(), src.cpu()), rep


def test_failed_blob_allocation_on_slot_k_releases_the_earlier_slots(monkeypatch):
    """raisr_hip_stream_set_model over several device slots allocates one staging blob per slot; when slot k cannot get one, the call
    returns ENOMEM and slots 0..k-1 are freed again (fault injection of the test-hooks flavour: RAISR_HIP_TEST_FAIL_BLOB_SLOT).  The ring
    stays usable: the same call without the fault succeeds and frames come out right."""
    import raisr_hip as R
    import synth
    L = R.lib_hooks()
    L.raisr_hip_debug_stream_live_blobs.restype = ctypes.c_int
    bank, qstr, qcoh, qa = R.read_model_folder(folder("filters_2x/filters_highres"), 8, 1)
    bank = np.ascontiguousarray(bank, np.float32); qstr = np.ascontiguousarray(qstr, np.float64); qcoh = np.ascontiguousarray(qcoh, np.float64)
    h = ctypes.c_void_p()
    devs = (ctypes.c_int * 3)(0, 0, 0)
    assert L.raisr_hip_stream_create_multi(ctypes.byref(h), devs, 3, 1) == 0
    try:
        args = (h, 0, bank.ctypes.data, bank.shape[0], bank.shape[1], qstr.ctypes.data, qcoh.ctypes.data, qa)
        assert L.raisr_hip_debug_stream_live_blobs() == 0
        for k in (0, 1, 2):
            monkeypatch.setenv("RAISR_HIP_TEST_FAIL_BLOB_SLOT", str(k))
            assert L.raisr_hip_stream_set_model(*args) == -3                                      # RAISR_HIP_ENOMEM (include/raisr_hip.h)
            assert L.raisr_hip_debug_stream_live_blobs() == 0, k
        monkeypatch.delenv("RAISR_HIP_TEST_FAIL_BLOB_SLOT")
        assert L.raisr_hip_stream_set_model(*args) == 0
        assert L.raisr_hip_debug_stream_live_blobs() == 0
    finally:
        L.raisr_hip_stream_destroy(h)
