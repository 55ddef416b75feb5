"""GPU box: self-check campaign of the certified hash stage at full C2 size.  Every pixel takes the approximate AND the exact
path; a certified bucket that differs from the exact one is counted (must stay 0).  Usage: certify_campaign.py [frames per kind]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "video-super-resolution-library_amd")]
from common import folder
import raisr_hip as R, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w, h = 1920, 1080
rng = np.random.default_rng(2024)
tot = {}
for asm, fold in ((2, "filters_2x/filters_highres"), (1, "filters_2x/filters_lowres"), (2, "filters_2x/filters_denoise")):
    dev = R.RaisrDevice(0)
    dev.set_model_from_folder(folder(fold), 8, 2)
    dev.configure(w, h, 2 * w, 2 * h, bits=8, passes=2, mode=1, hash_variant=asm)
    dev.certify_debug(True, True)
    out = np.zeros((2 * h, 2 * w), np.uint8)
    for i in range(n):
        kinds = [synth.natural_y(w, h, 8, seed=50000 + i), synth.random_y(w, h, 8, seed=60000 + i)]
        # smooth gradients with a little noise and hard edges: the content between "natural" and "constant"
        yy, xx = np.mgrid[0:h, 0:w]
        ph = rng.uniform(0, 6.28, 4); fr = rng.uniform(0.002, 0.05, 4)
        g = 128 + 60 * np.sin(fr[0] * xx + ph[0]) * np.cos(fr[1] * yy + ph[1]) + 30 * np.sin(fr[2] * (xx + yy) + ph[2])
        g[(xx * np.cos(ph[3]) + yy * np.sin(ph[3])) % 97 < 3] = 235
        kinds.append(np.clip(g + rng.integers(-1, 2, g.shape), 16, 235).astype(np.uint8))
        for y in kinds:
            dev.process_host(np.ascontiguousarray(y), out)
    st = dev.certify_stats()
    dev.close()
    key = f"asm{asm}_{fold.split('_')[-1]}"
    tot[key] = st
    print(key, json.dumps(st), "uncertified fraction %.4f" % (st["uncertain"] / st["pixels"]))
    assert st["mismatches"] == 0, st
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"frames_per_kind": n, "kinds": 3, "passes": 2, "results": tot}, open(os.path.join(ROOT, "gpurun_out", "certify_campaign.json"), "w"), indent=1)
