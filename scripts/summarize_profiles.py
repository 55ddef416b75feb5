#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (written by scripts/profile_gpu.sh) into profiles/<tag>_*.
usage: scripts/summarize_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
KERN = ("k_hash16", "k_filter16", "k_blend16", "k_hash", "k_filter", "k_blend", "k_resize2x", "k_resize")


def short(name):
    for k in KERN:
        if k + "<" in name or k + "(" in name or name.endswith(k):
            return k
    return name[:60]


lines = [f"# rocprofv3 summary `{tag}` — `python bench.py --no-cpu-baseline --steps 20 --warmup 3` (1080p->4K 2x, highres, 1-pass, 3 lanes x 24 frames/step)", ""]
f = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)   # newest run first
if f:
    lines += ["## --kernel-trace --stats", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f[0])):
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} |")
    lines.append("")
ev = os.path.join(src, "bench_events.json")
if os.path.exists(ev):
    try:
        j = json.loads(open(ev).read().strip().splitlines()[-1])
        lines += ["## bench.py HIP-event timing of the same command (un-profiled)", "", "```", json.dumps({k: j[k] for k in ("value", "unit", "ms_per_step", "kernels_avg_ms", "roofline")}, indent=1), "```", ""]
    except Exception as e:
        lines += [f"(bench_events.json unreadable: {e})", ""]
pmc = {}
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write", "pmc_l2"):
    f = sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("k_"):
            pmc.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
if pmc:
    lines += ["## PMC passes (`--lanes 1`, per launch averages; separate runs per counter group)", ""]
    cols = sorted({c for v in pmc.values() for c in v})
    lines += ["| kernel | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
    for k, v in pmc.items():
        lines.append(f"| {k} | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in cols) + " |")
    lines += ["", "Units: SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles summed over waves; FETCH_SIZE / WRITE_SIZE are KiB as reported.",
              "Calibration (MI355X_MICROARCH.md asks for one per access pattern): the gfx950 half-count of FETCH_SIZE applies to wide",
              "16-B/lane streams; these kernels load 1-4 B per lane.  Known byte counts: k_resize2x writes exactly one u16 LR plane",
              "(16 588 800 B = 16 200 KiB == WRITE_SIZE); k_blend must fetch the LR (u16) and HR (f32) planes with an 18/16 x 66/64 tile",
              "halo = 49.8 MB x 1.16 = 57.7 MB minimum vs FETCH_SIZE 62 980 KiB = 64.5 MB -- i.e. FETCH_SIZE is NOT halved for this",
              "access pattern, so no x2 correction is applied below.", ""]
    traffic = {}
    for k, v in pmc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic[k] = int((v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    if traffic:
        lines += ["## HBM bytes per launch (FETCH_SIZE + WRITE_SIZE) * 1024", "", "```", json.dumps(traffic, indent=1), "```", ""]
        json.dump({"k_hash_hbm_bytes_per_launch": traffic.get("k_hash"), "per_kernel": traffic,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (FETCH_SIZE + WRITE_SIZE)*1024; "
                             "no x2 FETCH correction: calibrated on k_resize2x/k_blend known byte counts (narrow per-lane loads), see the summary"},
                  open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
