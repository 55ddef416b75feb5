#!/bin/bash
# round 4, call 10: binary16 hash with ONE table look-up per VRCPPH(VRSQRTPH(.)) (composite table) -- parity + exhaustive checks, then C4 A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call10; mkdir -p $O
D=video-super-resolution-library_amd
( timeout 1200 python -m pytest tests/test_gpu_fold16.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_golden.py tests/test_gpu_batch.py -x -q -m gpu -k "fp16 or C4 or fold or golden or 1.5x" ) > $O/parity.log 2>&1; tail -2 $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
echo "== C4"
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config C4
run RAISR_HIP_FOLD16=1 --config C4
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config C4
run RAISR_HIP_FOLD16=1 --config C4
} 2>&1 | tee $O/ab.log
