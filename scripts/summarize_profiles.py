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
f = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
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
    f = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
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
    lines += ["", "Units: SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles summed over waves; FETCH_SIZE / WRITE_SIZE are KiB as reported",
              "(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide coalesced stream -> doubled below).", ""]
    traffic = {}
    for k, v in pmc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic[k] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    if traffic:
        lines += ["## HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE) * 1024", "", "```", json.dumps(traffic, indent=1), "```", ""]
        json.dump({"k_hash_hbm_bytes_per_launch": traffic.get("k_hash"), "per_kernel": traffic,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                             "(gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md HBM section)"},
                  open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
