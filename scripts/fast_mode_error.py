"""GPU box: error distribution of the fast mode (k_filter_mfma) against the oracle, per frame kind."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "video-super-resolution-library_amd")]
from common import folder, oracle_y, dtype_for
import raisr_hip as R, synth

case = ("2x_8b_avx512", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False)
w, h = 960, 540
rep = {}
for kind, y in (("natural", synth.natural_y(w, h, 8, seed=12345)), ("random", synth.random_y(w, h, 8, seed=777)),
                ("checker", synth.checker_y(w, h, 8)), ("constant", synth.constant_y(w, h, 8))):
    ref = oracle_y(y, case)
    outs = {}
    for fast in (0, 1, 2):
        dev = R.RaisrDevice(0)
        dev.set_model_from_folder(folder(case[1]), 8, 1)
        dev.configure(w, h, 2 * w, 2 * h, bits=8, passes=1, hash_variant=2)
        dev.set_fast(fast)
        out = np.zeros((2 * h, 2 * w), np.uint8)
        dev.process_host(y, out)
        outs[fast] = (out, dev.read_stage(0)[1])
        dev.close()
    assert np.array_equal(outs[0][0], ref)
    for level in (1, 2):
        d = np.abs(outs[level][0].astype(int) - ref.astype(int))
        hr_e, hr_f = outs[0][1], outs[level][1]
        z = (slice(6, 2 * h - 6), slice(6, 2 * w - 24))
        he = np.abs(hr_f[z] - hr_e[z])
        mse = float((d.astype(float) ** 2).mean())
        rep[f"{kind}_{level}"] = {"differ": float((d != 0).mean()), "hist": np.bincount(np.minimum(d.ravel(), 8), minlength=9).tolist(),
                     "max": int(d.max()), "psnr": None if mse == 0 else round(10 * np.log10(255 ** 2 / mse), 2),
                     "hr_err_mean": float(he.mean()), "hr_err_p999": float(np.quantile(he, 0.999)), "hr_err_max": float(he.max()),
                     "hr_big": int((he > 0.5).sum())}
        print(kind, level, json.dumps(rep[f"{kind}_{level}"]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "fast_mode_error.json"), "w"), indent=1)
