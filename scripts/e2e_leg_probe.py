"""GPU box: bench.py's end_to_end leg (synchronous RNLHandler_Process; page-locked and pageable planes) in a fresh process and after the pieces
of a bench run that precede it -- which of them costs the pageable path a third of its rate?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-super-resolution-library_amd"), os.path.join(ROOT, "oracle")]
sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch, raisr_hip as R
import numpy as np
wl = b.Workload("C2")


def e2e(tag):
    r = b.end_to_end_leg(R, wl, 256, 0)
    print("e2e", tag, r["fps"], r["pageable_planes"]["fps"], file=sys.stderr, flush=True)


def fence():
    torch.cuda.synchronize()


def blobs():
    out = []
    for p in range(wl.passes):
        bank, qs, qc, qa = R.read_model_folder(wl.folder, wl.bits, p + 1)
        out.append(torch.from_numpy(R.pack_model_blob(bank, qs, qc, qa)).cuda())
    return out


e2e("fresh")
bl = blobs()
frames = wl.frames("natural", range(8))
dt, kern, lanes, d_in, d_out = b.device_loop(R, torch, wl, 0, bl, 4, frames, 768, 2, 1, fence, False)
e2e("after a device loop without kernel timing (lanes alive)")
for d in lanes: d.close()
e2e("lanes closed")
dt, kern, lanes, d_in, d_out = b.device_loop(R, torch, wl, 0, bl, 4, frames, 768, 2, 1, fence, True)
e2e("after a device loop WITH kernel timing (lanes alive)")
iso = b.isolated_kernel_ms(lanes, d_in, d_out, wl, torch)
for d in lanes: d.close()
e2e("after isolated timing, lanes closed")
del d_in, d_out
torch.cuda.empty_cache()
e2e("device tensors freed")
