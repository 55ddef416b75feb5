"""GPU box: why does bench.py's end_to_end.pageable_planes read lower than scripts/e2e_probe.py?  The same synchronous RNLHandler_Process
loop with pageable planes after (a) nothing, (b) importing torch and touching the GPU, (c) a 4-lane device loop as bench.py runs first,
(d) the CPU oracle's OpenMP team having run (its threads spin for a while after a parallel region)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd"), os.path.join(ROOT, "oracle")]
import raisr_hip as R, synth

w, h = 1920, 1080
fold = os.path.join(ROOT, "filters_2x", "filters_highres")


def aligned(shape, fill=None):
    n = shape[0] * shape[1]
    raw = np.zeros(n + 4096, np.uint8)
    off = (-raw.ctypes.data) % 4096
    a = raw[off:off + n].reshape(shape)
    if fill is not None: a[...] = fill
    return a


def loop(tag, n=300, align=False):
    mk = (lambda a: aligned(a.shape, a)) if align else (lambda a: a)
    ys = [mk(synth.natural_y(w, h, 8, seed=i)) for i in range(4)]
    u = mk(synth.chroma(w // 2, h // 2, 8)); v = mk(u.copy())
    oy = mk(np.zeros((2 * h, 2 * w), np.uint8)); ou = mk(np.zeros((h, w), np.uint8)); ov = mk(np.zeros((h, w), np.uint8))
    tag += f" [oy at ...{oy.ctypes.data % 4096:04x}, y at ...{ys[0].ctypes.data % 4096:04x}]"
    R.RNLHandler_SetOpenCLContext(0, 0)
    assert R.RNLHandler_Init(fold, 2.0, 8, R.VideoRange, 20, R.HIP, 1, 1) == 0
    assert R.RNLHandler_SetRes((ys[0], u, v), (oy, ou, ov)) == 0
    for i in range(8): R.RNLHandler_Process((ys[i % 4], u, v), (oy, ou, ov))
    t0 = time.perf_counter()
    for i in range(n): R.RNLHandler_Process((ys[i % 4], u, v), (oy, ou, ov))
    dt = time.perf_counter() - t0
    R.RNLHandler_Deinit()
    print(f"{tag}: {n / dt:.0f} fps ({dt / n * 1e6:.0f} us/frame), threads in process: {len(os.listdir('/proc/self/task'))}", file=sys.stderr, flush=True)


loop("(a) fresh process")
import torch
torch.zeros(1).cuda(); torch.cuda.synchronize()
loop("(b) after importing torch and touching the GPU")
lanes = []
for _ in range(4):
    d = R.RaisrDevice(0); d.set_model_from_folder(fold, 8, 1); d.configure(w, h, 2 * w, 2 * h, bits=8, passes=1, hash_variant=R.HASH_AVX512); lanes.append(d)
d_in = [torch.from_numpy(synth.natural_y(w, h, 8, seed=i)).cuda() for i in range(8)]
d_out = [torch.empty((2 * h, 2 * w), dtype=torch.uint8, device="cuda") for _ in range(4)]
for f in range(2000):
    lanes[f % 4].process_y(d_in[f % 8].data_ptr(), w, d_out[f % 4].data_ptr(), 2 * w)
torch.cuda.synchronize()
loop("(c) after a 4-lane device loop (contexts still alive)")
for d in lanes: d.close()
loop("(c2) the same with the lanes closed")
import oracle_py as O
O.upscale_y(synth.natural_y(w, h, 8, seed=1), fold, 2.0, 8)
loop("(d) right after the OpenMP oracle ran")
time.sleep(2.0)
loop("(d2) two seconds later")
loop("(d3) again, planes 4 KB-aligned", align=True)
loop("(d4) again, plain planes")
