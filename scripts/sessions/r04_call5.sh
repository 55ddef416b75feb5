#!/bin/bash
# round 4, call 5: frame batches (raisr_hip_process_y_device_batch) -- bit-exactness, then frames per launch 1 / 2 / 4 / 8 / 16 on C1, C4, C2
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call5; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu ) > $O/batch.log 2>&1; tail -3 $O/batch.log
run() { echo -n "$1: "; python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 "${@:2}" 2>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_avg_ms'], d['kernels_isolated_ms'])" || tail -3 $O/err.log; }
{
for cfg in C1 C4 C2; do
  for b in 1 2 4 8 16 1 8; do run "$cfg batch $b" --config $cfg --batch $b; done
done
for cfg in C1 C4; do
  for l in 1 2 3; do run "$cfg batch 8 lanes $l" --config $cfg --batch 8 --lanes $l; done
done
} 2>&1 | tee $O/ab.log
