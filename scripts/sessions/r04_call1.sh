#!/bin/bash
# round 4, call 1: symmetric filter stage (filter_phase<.., SYM>) -- parity, then A/B against the eight-load stage on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_call1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_certify.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -3 $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
for cfg in C2 C5 C3; do
  echo "== $cfg"
  run RAISR_HIP_SYM=0 --config $cfg
  run RAISR_HIP_SYM=1 --config $cfg
  run "RAISR_HIP_SYM=1 RAISR_HIP_SYM_MAX_ROWS=64" --config $cfg
  run RAISR_HIP_SYM=0 --config $cfg
  run RAISR_HIP_SYM=1 --config $cfg
done 2>&1 | tee $O/ab.log
