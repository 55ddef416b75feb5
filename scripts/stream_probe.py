"""GPU box: where does the stream ring's time go?  fps vs ring depth, and host time inside submit() / collect()."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
import raisr_hip as R, synth

w, h = 1920, 1080
fold = os.path.join(ROOT, "filters_2x", "filters_highres")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for depth in (4, 4, 4, 4, 4, 4):
    pins = []
    def pinned(shape, fill=None):
        pl = R.PinnedPlane(shape, np.uint8); pins.append(pl)
        if fill is not None: pl.array[...] = fill
        return pl.array
    ys = [pinned((h, w), synth.natural_y(w, h, 8, seed=i)) for i in range(4)]
    u = pinned((h // 2, w // 2), synth.chroma(w // 2, h // 2, 8))
    frames = [R.PinnedFrame(2 * w, 2 * h, w, h, 8) for _ in range(depth)]
    pins.extend(frames)
    st = R.RaisrStream(0, fold, w, h, 2 * w, 2 * h, bits=8, chroma=(w // 2, h // 2, w, h), depth=depth)
    for warm in (True, False):
        m = 32 if warm else n
        ts = tc = 0.0
        t0 = time.perf_counter(); inflight = 0
        for i in range(m):
            if inflight == depth:
                a = time.perf_counter(); st.collect(); tc += time.perf_counter() - a; inflight -= 1
            f = frames[i % depth]
            a = time.perf_counter(); st.submit(ys[i % 4], u, u, f.y, f.u, f.v); ts += time.perf_counter() - a; inflight += 1
        while inflight:
            a = time.perf_counter(); st.collect(); tc += time.perf_counter() - a; inflight -= 1
        dt = time.perf_counter() - t0
    print(f"depth {depth:2d}: {n / dt:7.1f} fps   submit {ts / n * 1e6:6.1f} us/frame   collect {tc / n * 1e6:6.1f} us/frame   other {(dt - ts - tc) / n * 1e6:5.1f}")
    st.close()
    for p in pins: p.close()
