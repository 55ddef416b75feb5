#!/bin/bash
# round 5, call 20: two quarter-rate 64-bit multiply-adds out of the hot paths (bucket index in approx_hash, row base in filter_phase)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call20; mkdir -p $O
D=video-super-resolution-library_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_certify.py -q -x -m gpu 2>&1 | tail -3 | tee $O/tests.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
for cfg in C2 C5; do
echo "== $cfg natural"
for rep in 1 2 3 4; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run X=mad24 --config $cfg
done; done
} 2>&1 | tee $O/ab.log
