"""-m gpu: the streamed host pipeline (raisr_hip_stream_*): a ring of contexts with several frames in flight.  Every
frame of the stream must equal the oracle's output, in order, with page-locked and with pageable planes."""
import numpy as np
import pytest

from common import folder, oracle_y

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("bits,passes,asm", [(8, 1, 2), (10, 2, 2), (8, 1, 5)])
def test_stream_frames_equal_the_oracle_in_order(pinned, bits, passes, asm):
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h, depth, n = 160, 96, 3, 10
    fold = "filters_2x/filters_highres"
    dt = np.uint8 if bits == 8 else np.uint16
    case = ("x", fold, (2, 1), bits, passes, 1, asm, False)
    pins = []

    def plane(shape, fill=None):
        if pinned:
            pl = R.PinnedPlane(shape, dt); pins.append(pl); a = pl.array
        else:
            a = np.zeros(shape, dt)
        if fill is not None:
            a[...] = fill
        return a
    ys = [synth.natural_y(w, h, bits, seed=100 + i) for i in range(n)]
    refs = [oracle_y(y, case) for y in ys]
    uin = synth.random_y(w // 2, h // 2, bits, seed=5).astype(dt)
    uref = O.resize(uin, w, h).astype(dt)
    yin = [plane((h, w), y) for y in ys[:depth]]
    u = plane((h // 2, w // 2), uin)
    outs = [(plane((2 * h, 2 * w)), plane((h, w)), plane((h, w))) for _ in range(depth)]
    st = R.RaisrStream(0, folder(fold), w, h, 2 * w, 2 * h, bits=bits, passes=passes, hash_variant=asm,
                       chroma=(w // 2, h // 2, w, h), depth=depth)
    try:
        assert st.in_flight() == 0
        with pytest.raises(RuntimeError):
            st.collect()                                   # nothing in flight
        got = []
        inflight = 0
        for i in range(n):
            if inflight == depth:
                with pytest.raises(RuntimeError):
                    st.submit(yin[0], u, u, *outs[0])      # ring full: collect first
                st.collect(); inflight -= 1
                k = len(got)
                got.append((outs[k % depth][0].copy(), outs[k % depth][1].copy()))
            yin[i % depth][...] = ys[i]                    # the lane's previous frame has been collected: its planes are free
            st.submit(yin[i % depth], u, u, *outs[i % depth])
            inflight += 1
        while inflight:
            st.collect(); inflight -= 1
            k = len(got)
            got.append((outs[k % depth][0].copy(), outs[k % depth][1].copy()))
    finally:
        st.close()
        for pl in pins:
            pl.close()
    assert len(got) == n
    for i, (oy, ou) in enumerate(got):
        assert np.array_equal(oy, refs[i]), i
        assert np.array_equal(ou, uref), i


def test_packed_output_frame_is_downloaded_in_one_copy_and_matches():
    """Output planes laid out as raisr_hip_packed_frame_layout: same bits as separate planes."""
    import oracle_py as O
    import raisr_hip as R
    import synth
    w, h = 144, 80
    fold = "filters_2x/filters_highres"
    case = ("x", fold, (2, 1), 8, 1, 1, 2, False)
    y = synth.natural_y(w, h, 8, seed=77)
    u = synth.random_y(w // 2, h // 2, 8, seed=6)
    fo = R.PinnedFrame(2 * w, 2 * h, w, h, 8)
    st = R.RaisrStream(0, folder(fold), w, h, 2 * w, 2 * h, bits=8, chroma=(w // 2, h // 2, w, h), depth=2)
    try:
        for _ in range(3):
            fo.y[...] = 0; fo.u[...] = 0; fo.v[...] = 0
            st.submit(y, u, u, fo.y, fo.u, fo.v)
            st.collect()
            assert np.array_equal(fo.y, oracle_y(y, case))
            assert np.array_equal(fo.u, O.resize(u, w, h).astype(np.uint8)) and np.array_equal(fo.v, fo.u)
    finally:
        st.close()
        fo.close()


def test_stream_ring_from_a_device_blob_and_fast_refusal():
    """The multi-GPU start-up of a streamed rank: lanes take the packed blob from device memory (what the RCCL broadcast
    delivers) instead of reading files; same bits.  set_fast is refused while frames are in flight -- and, for a level > 0, always."""
    import torch
    import raisr_hip as R
    import synth
    from common import oracle_y
    fold = "filters_2x/filters_highres"
    w, h = 192, 108
    bank, qstr, qcoh, qa = R.read_model_folder(folder(fold), 8, 1)
    blob = torch.from_numpy(R.pack_model_blob(bank, qstr, qcoh, qa)).cuda()
    st = R.RaisrStream(0, "/nonexistent", w, h, 2 * w, 2 * h, bits=8, depth=2, blobs=[(blob.data_ptr(), blob.numel())])
    try:
        ys = [synth.natural_y(w, h, 8, seed=70 + i) for i in range(3)]
        outs = [np.zeros((2 * h, 2 * w), np.uint8) for _ in ys]
        for y, o in zip(ys[:2], outs[:2]):
            st.submit(y, None, None, o, None, None)
        with pytest.raises(RuntimeError):
            st.set_fast(1)                       # frames in flight
        st.collect(); st.collect()
        st.submit(ys[2], None, None, outs[2], None, None); st.collect()
        case = ("x", fold, (2, 1), 8, 1, 1, 2, False)
        for y, o in zip(ys, outs):
            assert np.array_equal(o, oracle_y(y, case))
        with pytest.raises(RuntimeError):
            st.set_fast(1)                       # the product library has no fast mode (round 4; development builds keep it)
        st.set_fast(0)
        st.submit(ys[0], None, None, outs[0], None, None); st.collect()
        assert np.array_equal(outs[0], oracle_y(ys[0], case))
    finally:
        st.close()


def test_context_on_caller_owned_streams():
    """raisr_hip_use_streams: host-plane frames on caller-owned streams (what the ring is built from) -- one stream for all three
    roles, and separate upload / compute / download streams with device-side events between the stages; same bits; restorable."""
    import torch
    import raisr_hip as R
    import synth
    fold = "filters_2x/filters_highres"
    w, h = 200, 120
    case = ("x", fold, (2, 1), 8, 2, 1, 2, False)
    y = synth.natural_y(w, h, 8, seed=31)
    u = synth.random_y(w // 2, h // 2, 8, seed=32)
    ref = oracle_y(y, case)
    s = [torch.cuda.Stream() for _ in range(3)]
    dev = R.RaisrDevice(0)
    try:
        dev.set_model_from_folder(folder(fold), 8, 2)
        dev.configure(w, h, 2 * w, 2 * h, bits=8, passes=2, mode=1, hash_variant=2)
        for streams in ((s[0], s[0], s[0]), (s[0], s[1], s[2]), None):
            if streams:
                dev.use_streams(*[x.cuda_stream for x in streams])
            else:
                dev.use_streams()
            oy = np.zeros((2 * h, 2 * w), np.uint8); ou = np.zeros((h, w), np.uint8); ov = np.zeros((h, w), np.uint8)
            dev.process_host(y, oy, u, ou, u, ov)
            assert np.array_equal(oy, ref) and np.array_equal(ou, ov) and ou.any()
        with pytest.raises(RuntimeError):
            dev.use_streams(s[0].cuda_stream, None, None)
    finally:
        dev.close()
