#!/usr/bin/env python3
"""usage: scripts/pmc_summarize.py <dir with g*/…/*_counter_collection.csv>  -- per-kernel averages of every counter"""
import collections, csv, glob, os, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "g*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_\w+", r["Kernel_Name"])
        if m:
            agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {sum(x)/len(x):14.5g}   (n={len(x)})")
