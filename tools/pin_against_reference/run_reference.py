#!/usr/bin/env python3
"""Runs the UNMODIFIED reference library (built by build_reference.sh, with real Intel IPP) on this repository's test frames and
writes tests/golden/reference_digests.json -- the file that turns "parity unpinned" into "pinned":

  * every case of tests/common.py:CASES on the two 96x64 frames of tests/golden/oracle_digests.json,
  * the five BASELINE.json configurations at full size on the frame bench.py runs (synth.natural_y, seed 12345),

each through RNLHandler_Init / SetRes / Process / Deinit with threadcount = 1 (whole-frame semantics), for the strict
build (normative) and the as-shipped -ffast-math build; sha256 of the Y output of both, the number of pixels in which they differ
(SURVEY s7 hard part 1 measured 83 of 8.29 M at 4K), and the number of pixels in which this repository's CPU oracle differs from the
strict build under either tie rule of the cheap upscale (RAISR_HIP_TIE_HALF_UP / _HALF_EVEN: Intel IPP's ippiResizeLinear is closed
source; which rule -- if either -- reproduces it is exactly what cannot be established without IPP).

Each case runs in its own process: the reference is a process-global singleton (Raisr_globals.h:140-203).

usage: run_reference.py --ref <reference checkout> --libs <dir with libraisr_ref_{strict,shipped}.so> [--out tests/golden/reference_digests.json]
Nothing of the reference is stored: digests and counts only."""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "video-super-resolution-library_amd")]


class VideoDataType(ctypes.Structure):          # Library/RaisrDefaults.h:56-62
    _fields_ = [("pData", ctypes.c_void_p), ("width", ctypes.c_uint), ("height", ctypes.c_uint), ("step", ctypes.c_uint), ("bitShift", ctypes.c_uint)]


def vdt(a):
    v = VideoDataType()
    v.pData, (v.height, v.width), v.step, v.bitShift = a.ctypes.data, a.shape, a.strides[0], 0
    return v


def worker(lib_path, ref_root, spec_path, out_path):
    """one case in this process: spec = {folder, ratio, bits, full, passes, mode, asm, in: npy path}"""
    spec = json.load(open(spec_path))
    y = np.load(spec["in"])
    L = ctypes.CDLL(lib_path)
    vp = ctypes.POINTER(VideoDataType)
    L.RNLHandler_Init.argtypes = [ctypes.c_char_p, ctypes.c_float, ctypes.c_uint, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
    L.RNLHandler_SetRes.argtypes = [vp] * 6
    L.RNLHandler_Process.argtypes = [vp] * 6 + [ctypes.c_int]
    rn, rd = spec["ratio"]
    h, w = y.shape
    ow, oh = w * rn // rd, h * rn // rd
    dt = y.dtype
    mid = 128 if spec["bits"] == 8 else 512
    u = np.full((h // 2, w // 2), mid, dt); v = u.copy()
    oy = np.zeros((oh, ow), dt); ou = np.zeros((oh // 2, ow // 2), dt); ov = ou.copy()
    os.chdir(ref_root)                                           # the reference resolves relative model folders against the cwd
    rc = L.RNLHandler_Init(spec["folder"].encode(), ctypes.c_float(rn / rd), spec["bits"], 2 if spec["full"] else 1, 1, spec["asm"],
                           spec["passes"], spec["mode"])          # RangeType: VideoRange = 1, FullRange = 2 (RaisrDefaults.h:54-57)
    if rc != 0:
        raise SystemExit(f"RNLHandler_Init failed: {rc:#x}")
    d = [vdt(a) for a in (y, u, v, oy, ou, ov)]
    if L.RNLHandler_SetRes(*[ctypes.byref(x) for x in d]) != 0:
        raise SystemExit("RNLHandler_SetRes failed")
    if L.RNLHandler_Process(*[ctypes.byref(x) for x in d], 2) != 0:          # CountOfBitsChanged
        raise SystemExit("RNLHandler_Process failed")
    L.RNLHandler_Deinit()
    np.save(out_path, oy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", required=True)
    ap.add_argument("--libs", required=True)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "reference_digests.json"))
    ap.add_argument("--only", default="", help="regular expression: run only the cases whose id matches (dry runs)")
    ap.add_argument("--worker", nargs=4, metavar=("LIB", "REF", "SPEC", "OUT"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.worker:
        return worker(*args.worker)

    import synth
    from common import CASES, oracle_y
    from make_golden import BASELINE, baseline_frame

    flags = open("/proc/cpuinfo").read()
    has_fp16 = "avx512_fp16" in flags
    if "GenuineIntel" not in flags or "avx512f" not in flags:
        print("WARNING: not an Intel AVX-512 host -- VRCP14PS / VRSQRT14PS results are vendor-specific; the digests written here are "
              "not the reference's bits on the hardware it targets", file=sys.stderr)
    # the reference's own model files must be the ones this repository ships
    model_files_identical = True
    import glob
    for top in glob.glob(os.path.join(ROOT, "filters_*")):
        for dirpath, _, files in os.walk(top):
            for f in files:
                mine = os.path.join(dirpath, f)
                theirs = os.path.join(args.ref, os.path.relpath(mine, ROOT))
                if not os.path.exists(theirs) or open(mine, "rb").read() != open(theirs, "rb").read():
                    model_files_identical = False

    jobs = []          # (id, case tuple, input frame)
    for case in CASES:
        bits = case[3]
        for nm, fr in (("natural", synth.natural_y(96, 64, bits, seed=4242)), ("random", synth.random_y(96, 64, bits, seed=99))):
            jobs.append((f"{case[0]}/{nm}", case, fr))
    for name in sorted(BASELINE):
        jobs.append((f"{name}/full", BASELINE[name][0], baseline_frame(name)))

    if args.only:
        import re
        jobs = [j for j in jobs if re.search(args.only, j[0])]
    cases = {}
    tie_votes = {"half_up": 0, "half_even": 0}
    with tempfile.TemporaryDirectory() as tmp:
        for jid, case, y in jobs:
            _, fold, ratio, bits, passes, mode, asm, full = case
            if asm == 5 and not has_fp16:
                cases[jid] = {"skipped": "this CPU lacks avx512_fp16: the reference would silently run its AVX-512 fp32 path (Raisr.cpp:1481-1512)"}
                continue
            np.save(os.path.join(tmp, "in.npy"), y)
            json.dump({"folder": fold, "ratio": list(ratio), "bits": bits, "full": full, "passes": passes, "mode": mode, "asm": asm,
                       "in": os.path.join(tmp, "in.npy")}, open(os.path.join(tmp, "spec.json"), "w"))
            outs = {}
            for build in ("strict", "shipped"):
                lib = os.path.join(args.libs, f"libraisr_ref_{build}.so")
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "--ref", args.ref, "--libs", args.libs,
                                       "--worker", lib, args.ref, os.path.join(tmp, "spec.json"), os.path.join(tmp, f"{build}.npy")],
                                      stdout=subprocess.DEVNULL)
                outs[build] = np.load(os.path.join(tmp, f"{build}.npy"))
            rec = {"in_shape": list(y.shape), "in_sha256": hashlib.sha256(y.tobytes()).hexdigest(),
                   "strict_sha256": hashlib.sha256(outs["strict"].tobytes()).hexdigest(),
                   "shipped_sha256": hashlib.sha256(outs["shipped"].tobytes()).hexdigest(),
                   "shipped_vs_strict_px": int((outs["strict"] != outs["shipped"]).sum()), "pixels": int(outs["strict"].size),
                   "oracle_vs_strict_px": {}}
            for tie_name, tie in (("half_up", 0), ("half_even", 1)):
                o = oracle_y(y, case, tie)
                bad = int((o != outs["strict"]).sum())
                rec["oracle_vs_strict_px"][tie_name] = bad
                if bad == 0:
                    tie_votes[tie_name] += 1
            cases[jid] = rec
            print(jid, rec["oracle_vs_strict_px"], "shipped differs in", rec["shipped_vs_strict_px"], flush=True)

    ran = [c for c in cases.values() if "skipped" not in c]
    tie = "half_up" if tie_votes["half_up"] == len(ran) else ("half_even" if tie_votes["half_even"] == len(ran) else "neither")
    info = open(os.path.join(args.libs, "build_info.txt")).read() if os.path.exists(os.path.join(args.libs, "build_info.txt")) else ""
    json.dump({"schema": 1, "generated_by": "tools/pin_against_reference/run_reference.py", "build_info": info.strip().splitlines(),
               "model_files_identical_to_this_repository": model_files_identical,
               "tie_rule_that_reproduces_ipp": tie, "cases_bit_exact": {k: v for k, v in tie_votes.items()}, "cases_run": len(ran),
               "cases": cases}, open(args.out, "w"), indent=1, sort_keys=True)
    print(f"wrote {args.out}: tie rule {tie}; oracle bit-exact on {max(tie_votes.values())} of {len(ran)} cases")


if __name__ == "__main__":
    main()
