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
import re


def short(name):
    m = re.search(r"\bk_\w+", name)
    return m.group(0) if m else name[:60]


variant = open(os.path.join(src, "variant.txt")).read().strip() if os.path.exists(os.path.join(src, "variant.txt")) else ""
m = re.search(r"--config\s+(C\d\w?)", variant)
config = m.group(1) if m else "C2"
variant_env = " ".join(t for t in variant.split() if "=" in t and not t.startswith("--"))
DESC = {"C1": "540p->1080p 2x, lowres, 1-pass, AVX2 numerics", "C2": "1080p->4K 2x, highres, 1-pass", "C2b": "1080p->4K 2x, highres, 1-pass, 10-bit (the reference's published configuration)", "C3": "1080p->4K 2x, highres, 2-pass",
        "C4": "720p->1080p 1.5x, denoise, 2-pass mode 2, binary16 numerics", "C5": "4K->8K 2x, highres, 10-bit"}
lines = [f"# rocprofv3 summary `{tag}` — `python bench.py --no-cpu-baseline --no-extras --config {config} --steps 3 --warmup 1` ({config}: {DESC[config]}, 4 lanes x 768 frames/step)"
         + (f", environment: `{variant_env}`" if variant_env else ""), ""]
f = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)   # newest run first
if f:
    lines += ["## --kernel-trace --stats", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f[0])):
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} |")
    lines.append("")
    import shutil
    shutil.copyfile(f[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))      # the raw rocprofv3 --stats table
ev = os.path.join(src, "bench_events.json")
if os.path.exists(ev):
    try:
        j = json.loads(open(ev).read().strip().splitlines()[-1])
        lines += ["## bench.py HIP-event timing of the same command (un-profiled)", "", "```", json.dumps({k: j[k] for k in ("value", "unit", "ms_per_step", "kernels_avg_ms", "roofline")}, indent=1), "```", ""]
    except Exception as e:
        lines += [f"(bench_events.json unreadable: {e})", ""]
pmc = {}
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write", "pmc_l2", "pmc_mfma"):
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
              "Calibration on known byte counts of these kernels' own access patterns (MI355X_MICROARCH.md asks for one):",
              "* WRITE_SIZE: k_resize2x writes exactly one u8 LR plane, 8 294 400 B = 8 100 KiB == WRITE_SIZE -> no correction.",
              "* FETCH_SIZE: k_blend4 (k_blend before round 6) has to fetch the LR plane (u8, 8.29 MB) and the HR plane (f32, 33.18 MB) that the previous",
              "  kernels wrote -- 41.5 MB per launch at the very least (the 4 MiB L2s cannot hold them; Infinity-Cache hits are",
              "  counted) -- and FETCH_SIZE reports half of that: the gfx950 half-count (128-B requests tallied at 64 B) applies to",
              "  these row-coalesced loads too, so FETCH_SIZE is doubled below, as the guide prescribes.", ""]
    # derived: VALU issue rate and LDS conflict share, per kernel, from the isolated (--lanes 1) durations of bench.py
    iso = {}
    fpl = 1                                                   # frames per launch of the profiled command (bench.py batches outputs <= 1080p by 8)
    try:
        jj = json.loads(open(ev).read().strip().splitlines()[-1])
        iso = jj.get("kernels_isolated_ms", {})
        fpl = int(jj.get("config", {}).get("frames_per_launch", 1)) or 1
    except Exception:
        pass
    alias = {"k_resize2x": "k_resize", "k_resize3x2": "k_resize", "k_blend4": "k_blend", "k_blend4_16": "k_blend16"}
    rows = []
    for k, v in pmc.items():
        t_ms = iso.get(alias.get(k, k))
        if t_ms and "SQ_INSTS_VALU" in v:
            # the counters are per LAUNCH (fpl frames), the isolated time is per FRAME: bring both to one frame (the round-4 summaries
            # of C1 / C4 divided 8-frame counters by a single-frame time: "341 %" of the probe rate)
            rate = v["SQ_INSTS_VALU"] / fpl / (t_ms * 1e-3) / 1e9
            conf = 100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0.0
            l2 = 100.0 * v.get("TCC_HIT_sum", 0.0) / (v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 1.0))
            rows.append(f"| {k} | {t_ms * 1e3:.1f} | {v['SQ_INSTS_VALU'] / fpl / 1e6:.1f} | {rate:.0f} | {100 * rate / 911:.0f} % | {conf:.1f} % | {l2:.1f} % |")
    if rows:
        lines += [f"## Derived (per frame: counters of a {fpl}-frame launch / {fpl}, isolated single-frame launch time, one lane)", "",
                  "| kernel | isolated us | VALU wave-inst (M) | G wave-inst/s | of the 911 G/s `v_fma_f32` probe rate | LDS cycles lost to bank conflicts | L2 hit rate |",
                  "|---|---|---|---|---|---|---|"] + rows + ["",
                  "(packed-fp32 instructions count once here but occupy two issue slots, so the tensor-heavy kernel's slot utilisation is",
                  "higher than its instruction rate suggests: see DESIGN.md s5.)", ""]
    traffic = {}
    for k, v in pmc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic[k] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    if traffic:
        lines += ["## HBM-side bytes per launch (2 * FETCH_SIZE + WRITE_SIZE) * 1024", "", "```", json.dumps(traffic, indent=1), "```", ""]
        dom = next((k for k in ("k_hashfilter_ac", "k_hashfilter", "k_hashfilter16", "k_filter_lds16", "k_hash") if k in traffic), None)
        import hashlib
        hs = hashlib.sha256()
        cs = os.path.join(root, "video-super-resolution-library_amd", "csrc")
        for fn in sorted(os.listdir(cs)):                      # same digest as bench.py source_hash(): the figure is only quoted for these sources
            if fn.endswith((".hip", ".h")) and fn != "host_copy.h":
                hs.update(fn.encode()); hs.update(open(os.path.join(cs, fn), "rb").read())
        # what the dominant kernel keeps busy (bench.py's roofline.binding reads this instead of typed-in figures): vector-instruction
        # rate against the measured v_fma_f32 rate of the part, LDS-busy share of the CU-cycles of the isolated launch, share of the wave
        # cycles spent waiting for an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)
        binding = None
        v = pmc.get(dom, {})
        t_ms = iso.get(alias.get(dom, dom)) if dom else None
        if t_ms and "SQ_INSTS_VALU" in v:
            CLOCK_HZ, N_CUS, PROBE_G = 2.4e9, 256, 911.0            # bench.py's constants; scripts/valu_rate_probe.hip (profiles/valu_rate_probe.json when recorded)
            try:
                PROBE_G = float(json.load(open(os.path.join(dst, "valu_rate_probe.json")))["v_fma_f32_G_wave_inst_per_s"])
            except Exception:
                pass
            rate = v["SQ_INSTS_VALU"] / fpl / (t_ms * 1e-3) / 1e9
            binding = {"kernel": dom, "isolated_us_per_frame": round(t_ms * 1e3, 1),
                       "valu_G_wave_inst_per_s": round(rate, 1), "valu_of_probe_rate": round(rate / PROBE_G, 3),
                       "lds_busy": round(v["SQ_LDS_IDX_ACTIVE"] / fpl / (t_ms * 1e-3 * CLOCK_HZ * N_CUS), 3) if "SQ_LDS_IDX_ACTIVE" in v else None,
                       "lds_cycles_lost_to_bank_conflicts": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 3) if v.get("SQ_LDS_IDX_ACTIVE") else None,
                       "wave_cycles_waiting_for_an_instruction": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in v else None,
                       "l2_hit_rate": round(v.get("TCC_HIT_sum", 0.0) / (v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 1.0)), 3) if "TCC_HIT_sum" in v else None,
                       "from": f"profiles/{tag}_rocprof_summary.md (rocprofv3 --pmc passes, --lanes 1)"}
        json.dump({"dominant_kernel": dom, "dominant_kernel_hbm_bytes_per_launch": traffic.get(dom), "per_kernel_bytes": traffic, "binding": binding,
                   "source_sha256": hs.hexdigest(), "config": config, "variant_env": variant_env,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (--lanes 1); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
                             "gfx950 FETCH_SIZE half-count confirmed on k_blend4's known byte count, WRITE_SIZE exact on k_resize2x's, see the summary"},
                  open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
