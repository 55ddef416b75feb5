"""GPU box: the plugin API with frames in flight (RNLHandler_Submit / _Collect, the FFmpeg filter's async=N path): fps by depth."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
import raisr_hip as R, synth

w, h = 1920, 1080
fold = os.path.join(ROOT, "filters_2x", "filters_highres")
n = 600
HOSTALLOC = int(os.environ.get("HOSTALLOC", "1"))         # 1: frame buffers from RNLHandler_HostAlloc (the FFmpeg filter's pools); 0: numpy
keep = []


def plane(shape, fill=None):
    if HOSTALLOC:
        keep.append(R.HostPlane(shape, np.uint8))
        a = keep[-1].array
    else:
        a = np.zeros(shape, np.uint8)
    if fill is not None:
        a[...] = fill
    return a


ys = [plane((h, w), synth.natural_y(w, h, 8, seed=i)) for i in range(4)]
u = plane((h // 2, w // 2), synth.chroma(w // 2, h // 2, 8)); v = plane((h // 2, w // 2), u)
for depth in (1, 2, 3, 4):
    outs = [(plane((2 * h, 2 * w)), plane((h, w)), plane((h, w))) for _ in range(depth)]
    R.RNLHandler_SetOpenCLContext(0, 0)
    assert R.RNLHandler_Init(fold, 2.0, 8, R.VideoRange, 20, R.HIP, 1, 1) == 0
    assert R.RNLHandler_SetRes((ys[0], u, v), outs[0]) == 0
    assert R.RNLHandler_SetAsyncDepth(depth) == 0
    for warm in (True, False):
        m = 16 if warm else n
        t0 = time.perf_counter()
        for i in range(m):
            if R.RNLHandler_FramesInFlight() == depth:
                assert R.RNLHandler_Collect() == 0
            assert R.RNLHandler_Submit((ys[i % 4], u, v), outs[i % depth]) == 0
        while R.RNLHandler_FramesInFlight():
            assert R.RNLHandler_Collect() == 0
        dt = time.perf_counter() - t0
    R.RNLHandler_Deinit()
    print(f"hostalloc={HOSTALLOC} async depth {depth}: {n / dt:.0f} fps ({dt / n * 1e6:.0f} us/frame)", file=sys.stderr)
