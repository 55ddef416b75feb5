"""-m gpu: BASELINE.json's five configurations at FULL size -- HIP path vs the CPU oracle, bit-exact,
plus size-independent properties (constant frames, border policy, lane/stream determinism)."""
import hashlib
import numpy as np
import pytest

from common import folder, dtype_for, oracle_y

pytestmark = pytest.mark.gpu

# id, folder, ratio, bits, passes, mode, asm, full_range, (in_w, in_h)
BASELINE = [
    ("C1_540p_lowres_avx2", "filters_2x/filters_lowres", (2, 1), 8, 1, 1, 1, False, (960, 540)),
    ("C2_1080p_highres_1p", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False, (1920, 1080)),
    ("C3_1080p_highres_2p", "filters_2x/filters_highres", (2, 1), 8, 2, 1, 2, False, (1920, 1080)),
    ("C4_720p_1.5x_denoise_fp16_2p_m2", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 5, False, (1280, 720)),
    ("C5_4k_8k_10bit", "filters_2x/filters_highres", (2, 1), 10, 1, 1, 2, False, (3840, 2160)),
    # not a BASELINE.json config: the one the reference publishes its numbers on (docs/performance.md:8-14), bench.py's C2b leg
    ("C2b_1080p_highres_1p_10bit", "filters_2x/filters_highres", (2, 1), 10, 1, 1, 2, False, (1920, 1080)),
]


def _gpu(y, case, lanes=1, via_torch=False):
    import raisr_hip as R
    _, fold, (rn, rd), bits, passes, mode, asm, full = case[:8]
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    outs = []
    devs = []
    for _ in range(lanes):
        dev = R.RaisrDevice(0)
        dev.set_model_from_folder(folder(fold), bits, passes)
        dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
        devs.append(dev)
    if via_torch:
        import torch
        d_in = torch.from_numpy(np.ascontiguousarray(y)).cuda()
        d_outs = [torch.zeros((oh, ow), dtype=d_in.dtype, device="cuda") for _ in devs]
        bps = y.dtype.itemsize
        for dev, d_out in zip(devs, d_outs):                      # all lanes in flight at once
            dev.process_y(d_in.data_ptr(), w * bps, d_out.data_ptr(), ow * bps)
        torch.cuda.synchronize()
        outs = [t.cpu().numpy() for t in d_outs]
    else:
        for dev in devs:
            out = np.zeros((oh, ow), dtype_for(bits))
            dev.process_host(np.ascontiguousarray(y), out)
            outs.append(out)
    for dev in devs:
        dev.close()
    return outs


@pytest.mark.parametrize("case", BASELINE, ids=[c[0] for c in BASELINE])
def test_full_size_bit_exact(case):
    import synth
    w, h = case[8]
    bits = case[3]
    y = synth.natural_y(w, h, bits, seed=2024)
    ref = oracle_y(y, case[:8])
    got = _gpu(y, case)[0]
    bad = np.argwhere(ref != got)
    assert bad.size == 0, f"{case[0]}: {len(bad)} mismatching pixels (max |d|={np.abs(ref.astype(int) - got.astype(int)).max()}), first {bad[:5].tolist()}"


def test_full_size_adversarial_frames_c2():
    import synth
    case = BASELINE[1]
    for name in ("random", "checker"):
        y = synth.FRAME_KINDS[name](1920, 1080, 8)
        ref = oracle_y(y, case[:8])
        got = _gpu(y, case)[0]
        assert np.array_equal(ref, got), name


@pytest.mark.parametrize("case", [BASELINE[1], BASELINE[2], BASELINE[3]], ids=["C2", "C3", "C4"])
def test_properties_full_size(case):
    import synth
    w, h = case[8]
    rn, rd = case[2]
    # constant input -> constant output (flat-patch bucket, filters preserve DC only up to the accept test + blend)
    y = synth.constant_y(w, h, 8, 128)
    out = _gpu(y, case)[0]
    assert out.shape == (h * rn // rd, w * rn // rd)
    assert np.all(out[0] == 128) and np.all(out[:, 0] == 128)
    # lanes / streams: 3 contexts processing the same device-resident frame concurrently give identical bytes,
    # equal to the host-staged path
    yn = synth.natural_y(w, h, 8, seed=5)
    a = _gpu(yn, case, lanes=3, via_torch=True)
    b = _gpu(yn, case)[0]
    digs = {hashlib.sha256(o.tobytes()).hexdigest() for o in a} | {hashlib.sha256(b.tobytes()).hexdigest()}
    assert len(digs) == 1


def test_border_policy_full_size_c2():
    """row 0 / H-1 and col 0 / W-1 are the unclamped cheap upscale; everything outside the filtered
    zone is clip(LR) -- checked against the library's own resize kernel."""
    import raisr_hip as R
    import synth
    import torch
    y = synth.random_y(1920, 1080, 8, seed=3)
    case = BASELINE[1]
    out = _gpu(y, case)[0]
    dev = R.RaisrDevice(0)
    d_in = torch.from_numpy(y).cuda()
    d_lr = torch.zeros((2160, 3840), dtype=torch.uint8, device="cuda")
    dev.resize_plane(d_in.data_ptr(), 1920, 1080, 1920, d_lr.data_ptr(), 3840, 2160, 3840, 8)
    dev.synchronize()
    lr = d_lr.cpu().numpy()
    dev.close()
    clip = np.clip(lr, 16, 235)
    assert np.array_equal(out[0], lr[0]) and np.array_equal(out[-1], lr[-1])
    assert np.array_equal(out[:, 0], lr[:, 0]) and np.array_equal(out[:, -1], lr[:, -1])
    assert np.array_equal(out[1:6, 1:-1], clip[1:6, 1:-1]) and np.array_equal(out[-6:-1, 1:-1], clip[-6:-1, 1:-1])
    assert np.array_equal(out[1:-1, 1:6], clip[1:-1, 1:6]) and np.array_equal(out[1:-1, 3830:-1], clip[1:-1, 3830:-1])


def test_c5_600_frame_stream_every_frame_equals_the_oracle():
    """BASELINE config 5's defining property on one GPU: a 600-frame 4K->8K 10-bit stream through the streamed host
    pipeline (the per-rank loop of `bench.py --config C5 --stream`).  The stream cycles over 3 distinct synthetic frames;
    every one of the 600 outputs must equal the oracle's output for its input, in order."""
    import raisr_hip as R
    import synth
    case = BASELINE[4]
    w, h = case[8]
    uniq, n, depth = 3, 600, 3
    ys = [synth.natural_y(w, h, 10, seed=12345 + i) for i in range(uniq)]
    refs = [oracle_y(y, case[:8]) for y in ys]
    cw, ch = w // 2, h // 2
    c = synth.chroma(cw, ch, 10)
    pins = [R.PinnedPlane(y.shape, np.uint16) for y in ys] + [R.PinnedPlane(c.shape, np.uint16)]
    for pl, src in zip(pins, ys + [c]):
        pl.array[...] = src
    fout = [R.PinnedFrame(2 * w, 2 * h, 2 * cw, 2 * ch, 10) for _ in range(depth)]
    st = R.RaisrStream(0, folder(case[1]), w, h, 2 * w, 2 * h, bits=10, chroma=(cw, ch, 2 * cw, 2 * ch), depth=depth)
    bad = []
    try:
        done = inflight = 0

        def collect():
            nonlocal done, inflight
            st.collect()
            if not np.array_equal(fout[done % depth].y, refs[done % uniq]) or not np.all(fout[done % depth].u == 512):
                bad.append(done)
            fout[done % depth].y[0, :8] = 0                      # stale bytes must not survive into the lane's next frame
            done += 1; inflight -= 1
        for i in range(n):
            if inflight == depth:
                collect()
            f = fout[i % depth]
            st.submit(pins[i % uniq].array, pins[uniq].array, pins[uniq].array, f.y, f.u, f.v)
            inflight += 1
        while inflight:
            collect()
    finally:
        st.close()
        for p in pins + fout:
            p.close()
    assert done == n and not bad, bad[:10]
