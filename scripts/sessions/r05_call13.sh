#!/bin/bash
# round 5, call 13: separable tensor passes on the matrix cores (RAISR_HIP_SEP_MFMA) -- parity subset, A/B inside one library
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call13; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_certify.py -q -x -m gpu 2>&1 | tail -4 | tee $O/tests.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for kind in natural random; do
echo "== C2 $kind"
for rep in 1 2 3; do
run RAISR_HIP_SEP_MFMA=0 --config C2 --frame-kind $kind
run RAISR_HIP_SEP_MFMA=1 --config C2 --frame-kind $kind
done; done
for cfg in C1 C3 C5; do
echo "== $cfg natural"
for rep in 1 2; do
run RAISR_HIP_SEP_MFMA=0 --config $cfg
run RAISR_HIP_SEP_MFMA=1 --config $cfg
done; done
} 2>&1 | tee $O/ab.log
python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('frame_kinds')))" | tee $O/kinds.json
