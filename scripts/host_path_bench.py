#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in host API (RNLHandler_Process: host planes in, host planes out),
1080p -> 4K yuv420p 8-bit, 1-pass highres.  Reported in DESIGN.md; never the bench `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-super-resolution-library_amd"))
import numpy as np  # noqa: E402
import raisr_hip as R  # noqa: E402
import synth  # noqa: E402

w, h, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 100
pinned = "--pinned" in sys.argv          # caller-owned page-locked planes (what hipHostMalloc / hipHostRegister give)
y = synth.natural_y(w, h, 8)
u = synth.chroma(w // 2, h // 2, 8); v = u.copy()
oy = np.zeros((2 * h, 2 * w), np.uint8); ou = np.zeros((h, w), np.uint8); ov = np.zeros((h, w), np.uint8)
if pinned:
    import torch
    keep = []
    def pin(a):
        t = torch.from_numpy(a).pin_memory(); keep.append(t)
        return t.numpy()
    y, u, v, oy, ou, ov = (pin(a) for a in (y, u, v, oy, ou, ov))
assert R.RNLHandler_Init(os.path.join(ROOT, "filters_2x/filters_highres"), 2.0, 8, R.VideoRange, 20, R.HIP, 1, 1) == 0
assert R.RNLHandler_SetRes((y, u, v), (oy, ou, ov)) == 0
for _ in range(5):
    R.RNLHandler_Process((y, u, v), (oy, ou, ov))
t0 = time.perf_counter()
for _ in range(n):
    R.RNLHandler_Process((y, u, v), (oy, ou, ov))
dt = time.perf_counter() - t0
R.RNLHandler_Deinit()
print(f"host path: {n / dt:.1f} fps, {2 * w * 2 * h * n / dt / 1e6:.0f} MP/s out-Y, {dt / n * 1e3:.3f} ms/frame (sync, yuv420p, incl. PCIe, {'pinned' if pinned else 'pageable'} host planes)")
