#!/bin/bash
# Final GPU session of round 3 (after the host-plane change touched device_abi.hip): the driver's suite command, the bench line,
# C5 by lanes in flight (HR + LR planes of 4 lanes vs the 256 MB Infinity Cache), rocprofv3 kernel stats + PMC passes for every
# BASELINE configuration, smoke.
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${TAG}_final2; mkdir -p $O
gcc -shared -fPIC -o /tmp/abort_trace.so scripts/abort_trace.c
( time LD_PRELOAD=/tmp/abort_trace.so timeout 1800 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -4 $O/bench.err
for l in 1 2 4; do echo -n "C5 lanes=$l: "; timeout 300 python bench.py --config C5 --lanes $l --no-cpu-baseline --no-extras --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['value'], d['kernels_avg_ms'])"; done > $O/c5_lanes.txt 2>&1; cat $O/c5_lanes.txt
for c in C2 C3 C4 C5 C1; do timeout 900 scripts/profile_gpu.sh ${TAG}_$c --config $c > $O/prof_$c.log 2>&1; echo "profiled $c"; done
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
