#!/bin/bash
# quick GPU iteration loop: small parity check + per-kernel timings (1 lane) + throughput (4 lanes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "96x64 or randomness" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --steps 8 --warmup 2"
for l in 1 4; do $B --lanes $l "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes', d['config']['lanes'], 'fps', d['config']['fps'], 'MP/s', d['value'], d['kernels_avg_ms'])"; done
