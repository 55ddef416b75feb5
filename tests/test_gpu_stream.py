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
