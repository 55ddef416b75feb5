#!/bin/bash
# round 6, session 13: frames in flight and launch batch on the round-6 kernels (side kernels are leaner and fit beside the main kernel
# now), and what the per-kernel HIP events of the timed region cost.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call13; mkdir -p $O
run() {
  echo -n "$*: "
  python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'])"
}
for r in 1 2; do
  for l in 3 4 5 6 8; do run --config C2 --lanes $l; done
  run --config C2 --no-kernel-timing
  for l in 3 4 6; do run --config C4 --lanes $l; done
  for b in 4 8 16; do run --config C4 --batch $b; done
  run --config C4 --no-kernel-timing
  for l in 4 6; do run --config C5 --lanes $l; done
  for b in 4 8 16; do run --config C1 --batch $b; done
done 2>&1 | tee $O/sweep.log
