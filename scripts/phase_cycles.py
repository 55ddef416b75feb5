"""GPU box, DEVELOPMENT build (RAISR_HIP_LIB=.../_exp/libraisr_dev.so): wave-cycles per phase of k_hashfilter_ac (C4: k_hashfilter16) on one BASELINE configuration
(s_memtime at the phase boundaries of every wave: csrc/kernels_common.h g_phase_cycles).  usage: phase_cycles.py [C2] [lanes]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-super-resolution-library_amd")]
sys.argv = ["bench.py"] + sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch, raisr_hip as R
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
lanes_n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = b.Workload(cfg)
bl = []
for p in range(wl.passes):
    bank, qs, qc, qa = R.read_model_folder(wl.folder, wl.bits, p + 1)
    bl.append(torch.from_numpy(R.pack_model_blob(bank, qs, qc, qa)).cuda())
L = R.lib()
out = (ctypes.c_ulonglong * 8)()
frames = wl.frames("natural", range(8))
dt, kern, lanes, d_in, d_out = b.device_loop(R, torch, wl, 0, bl, lanes_n, frames, 256, 1, 1, torch.cuda.synchronize, False)
L.raisr_hip_dev_phase_stats(out)                      # clear what warm-up and set-up left
dt, kern, lanes2, d_in, d_out = b.device_loop(R, torch, wl, 0, bl, lanes_n, frames, 256, 1, 0, torch.cuda.synchronize, False)
L.raisr_hip_dev_phase_stats(out)
names = ["window staging + barrier", "gradient tile + barrier", "V pass + barrier + H pass", "approximate hash + certification", "barrier after the hash",
         "worklist (exact path) + barriers", "filter stage", "-"]
if wl.asm == 5:                                       # binary16 pipeline: k_hashfilter16's marks (csrc/kernels_fp16.h)
    names = ["window + table staging + barrier", "gradient tile + barrier", "binary16 structure tensor", "per-pixel hash", "two barriers + pair windows", "-", "filter stage", "-"]
barrier = out[7]
out[7] = 0                                            # slot 7 = wave-cycles at the workgroup barriers: "of which", not a phase
tot = sum(out)
print(f"{cfg}, {lanes_n} lanes, {256 / dt:.0f} fps (instrumented); wave-cycles per phase, share of a wave's life:")
for n, v in zip(names, out):
    if v: print(f"  {n:38s} {v / tot * 100:5.1f} %   {v * 61 / (256 * wl.passes) / 1e6:8.2f} M wave-cycles per launch (sampled 1 workgroup in 61)")
if barrier:
    print(f"  of which parked at the workgroup barriers (arrival -> release, own outstanding memory operations included): {barrier / tot * 100:5.1f} %")
