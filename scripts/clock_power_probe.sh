#!/bin/bash
# (GPU box) shader clock and package power while the C2 device loop runs (4 lanes), against the idle figures: is the pipeline running at
# its power budget rather than at the 2.4 GHz the isolated-launch counters show?   usage: scripts/clock_power_probe.sh [bench.py args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/clock_power; mkdir -p $O
sample() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power|Average Graphics" | tr '\n' ' '; echo; }
echo "idle: $(sample)" | tee $O/log.txt
python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 40 --warmup 2 "$@" > $O/bench.json 2>/dev/null &
B=$!
sleep 4
for i in 1 2 3 4 5 6; do echo "load: $(sample)" | tee -a $O/log.txt; sleep 0.7; done
wait $B
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('fps', d['config']['fps'])" | tee -a $O/log.txt
