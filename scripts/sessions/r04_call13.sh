#!/bin/bash
# round 4, call 13: frames in flight (lanes) with the round-4 kernels; the default bench line; smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_call13; mkdir -p $O
run() { echo -n "$1: "; python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'])"; }
{
for cfg in C2 C3 C1 C4 C5; do for l in 2 3 4 6 8; do run "$cfg lanes $l" --config $cfg --lanes $l; done; done
} 2>&1 | tee $O/ab.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -4 $O/bench.err
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
