#!/usr/bin/env python3
"""usage: scripts/pmc_summarize.py <dir> [<dir> ...]  -- per-kernel averages of every counter found in *_counter_collection.csv below the directories"""
import collections, csv, glob, os, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
files = []
for d in sys.argv[1:]:
    files += glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
for f in files:
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_\w+", r["Kernel_Name"])
        if m:
            agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {sum(x)/len(x):14.5g}   (n={len(x)})")
