"""GPU box: the synchronous plugin path (RNLHandler_Process) with pageable vs page-locked caller planes and with RAISR_HIP_BANDS."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
import raisr_hip as R, synth

w, h = 1920, 1080
fold = os.path.join(ROOT, "filters_2x", "filters_highres")
n = int(os.environ.get("N", "400"))
pinned = int(os.environ.get("PIN", "0"))
hostalloc = int(os.environ.get("HOSTALLOC", "0"))          # 1: every plane from RNLHandler_HostAlloc
keep = []


def plane(a):
    if not hostalloc:
        return a
    keep.append(R.HostPlane(a.shape, a.dtype))
    keep[-1].array[...] = a
    return keep[-1].array


ys = [plane(synth.natural_y(w, h, 8, seed=i)) for i in range(4)]
u = plane(synth.chroma(w // 2, h // 2, 8)); v = plane(u.copy())
oy = plane(np.zeros((2 * h, 2 * w), np.uint8)); ou = plane(np.zeros((h, w), np.uint8)); ov = plane(np.zeros((h, w), np.uint8))
if pinned and not hostalloc:
    for a in ys + [u, v, oy, ou, ov]:
        assert R.lib().raisr_hip_host_register(a.ctypes.data, a.nbytes) == 0
R.RNLHandler_SetOpenCLContext(0, 0)
assert R.RNLHandler_Init(fold, 2.0, 8, R.VideoRange, 20, R.HIP, 1, 1) == 0
assert R.RNLHandler_SetRes((ys[0], u, v), (oy, ou, ov)) == 0
for i in range(8): R.RNLHandler_Process((ys[i % 4], u, v), (oy, ou, ov))
t0 = time.perf_counter()
for i in range(n): R.RNLHandler_Process((ys[i % 4], u, v), (oy, ou, ov))
dt = time.perf_counter() - t0
R.RNLHandler_Deinit()
print(f"hostalloc={hostalloc} pinned={pinned} bands={os.environ.get('RAISR_HIP_BANDS', '-')}: {n / dt:.0f} fps ({dt / n * 1e6:.0f} us/frame)", file=sys.stderr)
