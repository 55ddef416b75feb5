"""GPU box: self-check campaign of the certified hash stage at FULL frame size.  Every pixel takes the approximate AND the exact
path; a certified bucket that differs from the exact one is counted (must stay 0).
Round 3 matrix: 8-bit 2x (three models, both hash flavours, 2-pass mode 1 -- the round-2 campaign), plus 10-bit 2x, 16-bit 2x
(synthesised `_16` folder), 1.5x (one pixel type) and 2-pass mode 2 (pass 1 at input size), each on natural / noise /
smooth-gradient-with-edges frames.  Round 5 (KINDS=r05): three more frame families on the same matrix -- ramps (linear gradients of
integer and fractional slope in every orientation, piecewise: long runs of identical windows, exact symmetries), text-like edges
(two-level glyph strokes 1-3 px wide on flat or gently shaded ground) and film grain (natural frames + Gaussian grain).  Round 6 (KINDS=photo): real photographs (photos.py).
Usage: [KINDS=r05|photo] certify_campaign.py [frames per kind] -> gpurun_out/certify_campaign.json"""
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
    if os.environ.get("KINDS") == "photo":
        # round 6: REAL pictures (photos.py: the photographs installed with this image's Python packages; plain / JPEG-blocky / letterboxed /
        # enlarged / mosaic).  Three frames per i; 13 is coprime with sources x variants, so 35 values of i visit every (source, variant) pair.
        import photos
        if bits == 16:
            return [(photos.photo_y(w, h, 10, 13 * (3 * i + k)).astype(np.uint32) * 64 + rng.integers(0, 64, (h, w))).astype(np.uint16) for k in range(3)]
        return [photos.photo_y(w, h, bits, 13 * (3 * i + k)) for k in range(3)]
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
    if os.environ.get("KINDS") != "r05":
        return [nat, rnd, smooth]
    # ramps: piecewise linear gradients, slopes from 1/16 to 4 levels per pixel (scaled to the bit depth), random orientation per band
    sc = max(1, maxv // 255)
    ramp = np.zeros((h, w), np.float64)
    band = max(32, h // 12)
    for b0 in range(0, h, band):
        th = rng.uniform(0, np.pi)
        sl = rng.choice([1 / 16, 1 / 4, 1 / 2, 1, 2, 4]) * sc
        ramp[b0:b0 + band] = (lo + (sl * (xx[b0:b0 + band] * np.cos(th) + yy[b0:b0 + band] * np.sin(th))) % (hi - lo))
    ramp = np.floor(ramp).clip(lo, hi).astype(nat.dtype)
    # text-like: strokes (horizontal, vertical, diagonal) 1-3 px wide, dark on light or light on dark, on a flat or faintly shaded ground
    ground = (lo + (hi - lo) * (0.75 + 0.05 * np.sin(0.003 * xx + ph[0]))).astype(np.float64)
    ink = float(lo + (hi - lo) * 0.08)
    txt = ground.copy()
    for _ in range(w * h // 900):
        x0, y0 = int(rng.integers(0, w - 24)), int(rng.integers(0, h - 24))
        ln, wd, kind = int(rng.integers(5, 22)), int(rng.integers(1, 4)), int(rng.integers(0, 4))
        if kind == 0:
            txt[y0:y0 + wd, x0:x0 + ln] = ink
        elif kind == 1:
            txt[y0:y0 + ln, x0:x0 + wd] = ink
        else:
            for t in range(ln):
                xx0 = x0 + t if kind == 2 else x0 + ln - t
                txt[y0 + t:y0 + t + wd, xx0:xx0 + wd] = ink
    if i & 1:
        txt = lo + hi - txt                                   # light on dark
    txt = txt.clip(lo, hi).astype(nat.dtype)
    # film grain: the natural frame with Gaussian grain of sigma 1.5-6 levels (8-bit scale), brighter areas a little grainier
    sig = rng.uniform(1.5, 6.0) * sc
    grain = np.clip(nat.astype(np.float64) + rng.standard_normal((h, w)) * sig * (0.6 + 0.4 * nat / maxv), 0, maxv).astype(nat.dtype)
    return [ramp, txt, grain]


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
KIND_NAMES = {"r05": ["ramps", "text-like edges", "film grain"],
              "photo": ["photographs (photos.py), three frames per index: plain / jpeg30 / letterbox / soft2x / mosaic variants of the installed sample pictures"]}
json.dump({"frames_per_kind": n, "kinds": KIND_NAMES.get(os.environ.get("KINDS"), ["natural", "noise", "smooth gradients with hard edges"]),
           "buckets_compared": sum(v["pixels"] for v in tot.values()), "certified_but_wrong": sum(v["mismatches"] for v in tot.values()),
           "results": tot}, open(os.path.join(ROOT, "gpurun_out", "certify_campaign.json"), "w"), indent=1)
