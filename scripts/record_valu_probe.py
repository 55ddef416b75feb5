#!/usr/bin/env python3
"""Turns the output of scripts/valu_rate_probe.hip (a file, or stdin) into profiles/valu_rate_probe.json -- the sustained v_fma_f32 rate of the
part that bench.py's roofline.valu.sustained_peak and scripts/summarize_profiles.py's `binding` object quote.
usage (build container, after a GPU session wrote the probe's output):  scripts/record_valu_probe.py gpurun_out/<session>/valu_rate_probe.txt"""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "-"
text = sys.stdin.read() if src == "-" else open(src).read()
m = re.search(r"^v_fma_f32\s+([\d.]+) ms\s+([\d.]+) G wave-inst/s\s+([\d.]+) TFLOP/s", text, re.M)
if not m:
    sys.exit("no v_fma_f32 line in the probe's output")
out = {"v_fma_f32_G_wave_inst_per_s": float(m.group(2)), "v_fma_f32_TFLOP_per_s": float(m.group(3)),
       "from": f"scripts/valu_rate_probe.hip on one MI355X ({os.path.basename(src)})", "raw": text.strip().splitlines()}
json.dump(out, open(os.path.join(root, "profiles", "valu_rate_probe.json"), "w"), indent=1)
print(out)
