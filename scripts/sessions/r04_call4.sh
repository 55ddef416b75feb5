#!/bin/bash
# round 4, call 4: binary16 pipeline -- one tree per row, pair windows, folded thresholds: parity + exhaustive fold check, then C4 A/B
# against the previous library on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call4; mkdir -p $O
D=video-super-resolution-library_amd
( timeout 1200 python -m pytest tests/test_gpu_fold16.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipelines.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -3 $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
echo "== C4"
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config C4
run RAISR_HIP_FOLD16=1 --config C4
run RAISR_HIP_FOLD16=0 --config C4
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config C4
run RAISR_HIP_FOLD16=1 --config C4
run RAISR_HIP_FOLD16=0 --config C4
} 2>&1 | tee $O/ab.log
