"""GPU box: create -> set model -> configure -> first frame, back to back, many times (small 2-pass frames): does a configure-time
clear ever overtake the first frame?  Prints the number of iterations whose output differs from the first iteration's."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-super-resolution-library_amd")]
import raisr_hip as R, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
fold = os.path.join(ROOT, "filters_2x", "filters_denoise")
bad = 0; ref = None
for it in range(n):
    w, h = (96, 64) if it % 2 == 0 else (160, 90)
    y = synth.checker_y(w, h, 10) if it % 4 < 2 else synth.natural_y(w, h, 10, seed=5)
    dev = R.RaisrDevice(0)
    dev.set_model_from_folder(fold, 10, 2)
    dev.configure(w, h, 2 * w, 2 * h, bits=10, passes=2, mode=2, hash_variant=2)
    out = np.zeros((2 * h, 2 * w), np.uint16)
    dev.process_host(y, out)
    dev.close()
    key = it % 4
    if ref is None: ref = {}
    if key not in ref: ref[key] = out.copy()
    elif not np.array_equal(out, ref[key]): bad += 1
print(f"{bad} of {n} iterations differ")
