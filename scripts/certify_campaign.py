"""GPU box: self-check campaign of the certified hash stage at FULL frame size.  Every pixel takes the approximate AND the exact
path; a certified bucket that differs from the exact one is counted (must stay 0).
Round 3 matrix: 8-bit 2x (three models, both hash flavours, 2-pass mode 1 -- the round-2 campaign), plus 10-bit 2x, 16-bit 2x
(synthesised `_16` folder), 1.5x (one pixel type) and 2-pass mode 2 (pass 1 at input size), each on natural / noise /
smooth-gradient-with-edges frames.  Usage: certify_campaign.py [frames per kind] -> gpurun_out/certify_campaign.json"""
import json, os, shutil, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "video-super-resolution-library_amd")]
from common import folder
import raisr_hip as R, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2024)
tmp = tempfile.mkdtemp()
f16 = os.path.join(tmp, "filters16")
shutil.copytree(folder("filters_2x/filters_highres"), f16)
for stem in ("filterbin_2", "Qfactor_strbin_2", "Qfactor_cohbin_2"):
    for sfx in ("", "_2"):
        shutil.copyfile(os.path.join(f16, f"{stem}_10{sfx}"), os.path.join(f16, f"{stem}_16{sfx}"))

# (key, folder, in_w, in_h, out_w, out_h, bits, passes, mode, asm, full_range, frames per kind)
MATRIX = [
    ("8b_2x_avx512_highres_2p", folder("filters_2x/filters_highres"), 1920, 1080, 3840, 2160, 8, 2, 1, 2, False, n),
    ("8b_2x_avx2_lowres_2p", folder("filters_2x/filters_lowres"), 1920, 1080, 3840, 2160, 8, 2, 1, 1, False, n),
    ("8b_2x_avx512_denoise_2p", folder("filters_2x/filters_denoise"), 1920, 1080, 3840, 2160, 8, 2, 1, 2, False, n),
    ("10b_2x_avx512_highres_2p", folder("filters_2x/filters_highres"), 1920, 1080, 3840, 2160, 10, 2, 1, 2, False, max(4, n // 2)),
    ("10b_2x_avx2_denoise_2p_mode2_full", folder("filters_2x/filters_denoise"), 1920, 1080, 3840, 2160, 10, 2, 2, 1, True, max(4, n // 2)),
    ("16b_2x_avx512_highres_2p_full", f16, 1920, 1080, 3840, 2160, 16, 2, 1, 2, True, max(4, n // 2)),
    ("8b_1.5x_avx512_denoise_2p_mode2", folder("filters_1.5x/filters_denoise"), 1280, 720, 1920, 1080, 8, 2, 2, 2, False, n),
    ("8b_1.5x_avx2_highres_1p", folder("filters_1.5x/filters_highres"), 1280, 720, 1920, 1080, 8, 1, 1, 1, False, n),
    ("10b_4k_8k_avx512_highres_1p", folder("filters_2x/filters_highres"), 3840, 2160, 7680, 4320, 10, 1, 1, 2, False, max(2, n // 8)),
]


def frames(i, w, h, bits):
    maxv = (1 << bits) - 1
    lo, hi = (16 * maxv // 255, 235 * maxv // 255)
    if bits == 16:                              # 10-bit generators scaled up, with low-order noise so that all 16 bits are in play
        nat = (synth.natural_y(w, h, 10, seed=50000 + i).astype(np.uint32) * 64 + rng.integers(0, 64, (h, w))).astype(np.uint16)
        rnd = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    else:
        nat, rnd = synth.natural_y(w, h, bits, seed=50000 + i), synth.random_y(w, h, bits, seed=60000 + i)
    yy, xx = np.mgrid[0:h, 0:w]
    ph = rng.uniform(0, 6.28, 4); fr = rng.uniform(0.002, 0.05, 4)
    g = 0.5 + 0.235 * np.sin(fr[0] * xx + ph[0]) * np.cos(fr[1] * yy + ph[1]) + 0.118 * np.sin(fr[2] * (xx + yy) + ph[2])
    g = g * maxv
    g[(xx * np.cos(ph[3]) + yy * np.sin(ph[3])) % 97 < 3] = hi
    smooth = np.clip(g + rng.integers(-1, 2, g.shape) * max(1, maxv // 255), lo, hi).astype(nat.dtype)
    return [nat, rnd, smooth]


tot = {}
for key, fold, w, h, ow, oh, bits, passes, mode, asm, full, nn in MATRIX:
    dev = R.RaisrDevice(0, hooks=True)
    dev.set_model_from_folder(fold, bits, passes)
    dev.configure(w, h, ow, oh, bits=bits, full_range=full, passes=passes, mode=mode, hash_variant=asm)
    dev.certify_debug(True, True)
    out = np.zeros((oh, ow), np.uint8 if bits == 8 else np.uint16)
    for i in range(nn):
        for y in frames(i, w, h, bits):
            dev.process_host(np.ascontiguousarray(y), out)
    st = dev.certify_stats()
    # the 32-bit device counters wrap on long runs of big frames: keep the per-configuration count below 2^32 (checked here)
    assert st["pixels"] < (1 << 32)
    dev.close()
    st["frames"] = 3 * nn
    st["uncertified_frac"] = round(st["uncertain"] / st["pixels"], 6)
    tot[key] = st
    print(key, json.dumps(st), flush=True)
    assert st["mismatches"] == 0, (key, st)
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"frames_per_kind": n, "kinds": ["natural", "noise", "smooth gradients with hard edges"],
           "buckets_compared": sum(v["pixels"] for v in tot.values()), "certified_but_wrong": sum(v["mismatches"] for v in tot.values()),
           "results": tot}, open(os.path.join(ROOT, "gpurun_out", "certify_campaign.json"), "w"), indent=1)
