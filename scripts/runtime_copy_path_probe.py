"""GPU box: which path does the HIP runtime take for asynchronous copies on PAGEABLE host memory?  Run with AMD_LOG_LEVEL=4 and
RAISR_HIP_BOUNCE=0 (pageable planes handed to the runtime as in rounds 1-2); the caller greps the log for the runtime's own
'Pinned' / 'Unpinned' / 'Staged' messages.  usage: python scripts/runtime_copy_path_probe.py <strided 0|1> <w> <h>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
import raisr_hip as R, synth

strided, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pad = 10 if strided else 0


def mk(a):
    buf = np.zeros((a.shape[0], a.shape[1] + pad), a.dtype)
    buf[:, :a.shape[1]] = a
    return buf[:, :a.shape[1]]


y = synth.natural_y(w, h, 8, seed=1)
u = synth.chroma(w // 2, h // 2, 8)
R.RNLHandler_SetOpenCLContext(0, 0)
assert R.RNLHandler_Init(os.path.join(ROOT, "filters_2x", "filters_highres"), 2.0, 8, R.VideoRange, 20, R.HIP, 1, 1) == 0
outs = (mk(np.zeros((2 * h, 2 * w), np.uint8)), mk(np.zeros((h, w), np.uint8)), mk(np.zeros((h, w), np.uint8)))
ins = (mk(y), mk(u), mk(u))
assert R.RNLHandler_SetRes(ins, outs) == 0
sys.stderr.write("=== PROBE BEGIN\n"); sys.stderr.flush()
for _ in range(3):
    assert R.RNLHandler_Process(ins, outs) == 0
sys.stderr.write("=== PROBE END\n"); sys.stderr.flush()
R.RNLHandler_Deinit()
