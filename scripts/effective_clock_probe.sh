#!/bin/bash
# (GPU box) effective shader clock of the hot kernels under a sustained run: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration,
# first and last quarter of a run of ~2 000 back-to-back dispatches (rocprofv3 serialises dispatches while it collects counters).
# usage: scripts/effective_clock_probe.sh [bench.py args]   -> gpurun_out/effclock/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/effclock; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d $O/run -- python $R/bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 2 --warmup 1 --frames-per-step 256 "$@" > $O/run.log 2>&1
python - <<'P' | tee $O/summary.txt
import csv, glob, os, re, collections
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/effclock"
cc = glob.glob(O + "/run/**/*counter_collection.csv", recursive=True)
kt = glob.glob(O + "/run/**/*kernel_trace.csv", recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
rows = collections.defaultdict(list)
for f in cc:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
        s, e, name = dur[r["Dispatch_Id"]]
        m = re.search(r"\bk_\w+", name)
        if m: rows[m.group(0)].append((s, float(r["Counter_Value"]) / 8.0 / max(e - s, 1)))
for k, v in rows.items():
    v.sort(); n = len(v); q = max(n // 4, 1)
    first = sum(x for _, x in v[:q]) / q; last = sum(x for _, x in v[-q:]) / q
    print(f"{k}: {n} dispatches, effective clock first quarter {first:.3f} GHz, last quarter {last:.3f} GHz")
P
