#!/bin/bash
# round 5, call 27: does instruction fetch bind k_hashfilter_ac?  (27.4 KB of code, 16 waves per CU in different stages, two CUs per
# instruction cache, k_blend / k_resize2x of other frames beside it.)  Instruction-cache and fetch counters of the C2 pipeline, one lane and four.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r05_call27; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQ_INST_LEVEL|SQC_INST|SQ_INSTS_SMEM|SQC_DCACHE_MISSES|SQC_DCACHE_REQ " | head -40 > $O/avail.txt
for lanes in 1 4; do
B="python $R/bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 2 --warmup 1 --lanes $lanes --frames-per-step 8"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/ic_l$lanes -- $B > $O/ic_l$lanes.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/if_l$lanes -- $B > $O/if_l$lanes.log 2>&1
done
python - <<'P' | tee $O/summary.txt
import csv, glob, collections, os, re
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r05_call27"
for d in sorted(glob.glob(O + "/i[cf]_l*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"\bk_\w+", r["Kernel_Name"]); k = m.group(0) if m else r["Kernel_Name"][:30]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    print("==", os.path.basename(d))
    for k in acc:
        print(" ", k, {c: round(v / max(n[(k, c)], 1)) for c, v in acc[k].items()})
P
