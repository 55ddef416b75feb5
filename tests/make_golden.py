"""Regenerates the SELF-GENERATED regression fixtures under tests/golden/ from the CPU oracle (run from the repo root):

  oracle_digests.json    sha256 of the oracle's Y output for every case of tests/common.py:CASES on two seeded 96x64 frames
  baseline_digests.json  SURVEY s8(c)(iii): sha256 of the Y output of the five BASELINE.json configurations at FULL size (frame =
                         synth.natural_y, seed 12345 -- the frame bench.py runs), plus the strided checksum-of-rows used by
                         tests that want to say WHERE a frame differs
  stages_2x_highres_8b_96x64.npz
                         SURVEY s8(c)(ii): per-stage dumps of one case -- input, cheap upscale (LR u8), hash bucket per pixel
                         (0xFF outside the filtered zone), HR plane (fp32 bit patterns), blended output

They pin the oracle (and, under -m gpu, the HIP path) against drift.  They are NOT evidence about the reference, which ships no
golden vectors and cannot be built in this image (Intel IPP): see DESIGN.md s3 and tools/pin_against_reference/."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "video-super-resolution-library_amd")]
import oracle_py as O  # noqa: E402
import synth  # noqa: E402
from common import CASES, folder, dtype_for, oracle_y  # noqa: E402

# the five BASELINE.json configurations as tests/common.py-style cases: (id, folder, ratio, bits, passes, mode, asm, full), input size
BASELINE = {
    "C1": (("C1", "filters_2x/filters_lowres", (2, 1), 8, 1, 1, 1, False), (960, 540)),
    "C2": (("C2", "filters_2x/filters_highres", (2, 1), 8, 1, 1, 2, False), (1920, 1080)),
    "C3": (("C3", "filters_2x/filters_highres", (2, 1), 8, 2, 1, 2, False), (1920, 1080)),
    "C4": (("C4", "filters_1.5x/filters_denoise", (3, 2), 8, 2, 2, 5, False), (1280, 720)),
    "C5": (("C5", "filters_2x/filters_highres", (2, 1), 10, 1, 1, 2, False), (3840, 2160)),
}


def row_checksums(a):
    """adler32 of every 64th row: a few hundred bytes that localise a difference"""
    return [zlib.adler32(a[r].tobytes()) for r in range(0, a.shape[0], 64)]


def baseline_frame(name):
    case, (w, h) = BASELINE[name]
    return synth.natural_y(w, h, case[3], seed=12345)


def main():
    got = {}
    for cid, fold, (rn, rd), bits, passes, mode, asm, full in CASES:
        for nm, fr in (("natural", synth.natural_y(96, 64, bits, seed=4242)), ("random", synth.random_y(96, 64, bits, seed=99))):
            out = oracle_y(fr, (cid, fold, (rn, rd), bits, passes, mode, asm, full))
            got[f"{cid}/{nm}"] = hashlib.sha256(out.tobytes()).hexdigest()
    json.dump(got, open(os.path.join(HERE, "golden", "oracle_digests.json"), "w"), indent=1, sort_keys=True)
    print(len(got), "case digests written")

    base = {}
    for name, (case, (w, h)) in BASELINE.items():
        y = baseline_frame(name)
        out = oracle_y(y, case)
        base[name] = {"input": f"synth.natural_y({w}, {h}, {case[3]}, seed=12345)", "input_sha256": hashlib.sha256(y.tobytes()).hexdigest(),
                      "out_shape": list(out.shape), "out_dtype": str(out.dtype), "sha256": hashlib.sha256(out.tobytes()).hexdigest(),
                      "row_adler32_every_64": row_checksums(out)}
        print(name, base[name]["sha256"][:16])
    json.dump(base, open(os.path.join(HERE, "golden", "baseline_digests.json"), "w"), indent=1, sort_keys=True)

    # per-stage dumps of one case (2x highres, 8-bit, 1 pass, AVX-512 numerics, 96x64 -> 192x128)
    y = synth.natural_y(96, 64, 8, seed=4242)
    lr = O.resize(y, 192, 128)
    p = O.make_pass(O.Model(folder("filters_2x/filters_highres"), 8, 1), 8, False, O.ASM_AVX512)
    out, hd, hr = O.run_pass(lr, p, dumps=True)
    np.savez_compressed(os.path.join(HERE, "golden", "stages_2x_highres_8b_96x64.npz"), input=y, lr=lr.astype(np.uint8), hash=hd.astype(np.uint8),
                        hr_bits=np.ascontiguousarray(hr, np.float32).view(np.uint32), out=out.astype(np.uint8))
    print("stage dumps written")


if __name__ == "__main__":
    main()
