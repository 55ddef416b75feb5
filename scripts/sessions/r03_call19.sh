#!/bin/bash
# GPU session 19 of round 3: frames in flight per GPU (lanes) for the small-frame configurations (one launch of C4 / C1 is 1-2 rounds of workgroups)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call19; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['value'])"; }
{
for c in C4 C1 C2 C3; do for l in 2 4 6 8 12; do echo -n "$c lanes=$l: "; $B --config $c --lanes $l 2>/dev/null | show; done; done
} > $O/lanes.txt 2>&1
cat $O/lanes.txt
