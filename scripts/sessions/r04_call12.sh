#!/bin/bash
# round 4, call 12: on-the-fly 2x cheap upscale (no k_resize2x launch, no LR plane for pass 1) -- parity, host paths, then A/B on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call12; mkdir -p $O
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_batch.py tests/test_gpu_golden.py tests/test_gpu_bands.py tests/test_gpu_host_api.py tests/test_gpu_pipelines.py tests/test_gpu_certify.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -3 $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C3 C5 C1; do
echo "== $cfg"
run RAISR_HIP_UP2X=0 --config $cfg
run RAISR_HIP_UP2X=1 --config $cfg
run RAISR_HIP_UP2X=0 --config $cfg
run RAISR_HIP_UP2X=1 --config $cfg
done
} 2>&1 | tee $O/ab.log
