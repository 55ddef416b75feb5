#!/bin/bash
# round 5, call 12: coherence bound with the 1 / (1 + t)^2 factor -- parity + certify subset, uncertified share, A/B against the previous library
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call12; mkdir -p $O
D=video-super-resolution-library_amd
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_certify.py tests/test_gpu_hash_unit.py -q -x -m gpu 2>&1 | tail -4 | tee $O/tests.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for kind in natural random; do
echo "== C2 $kind"
for rep in 1 2 3; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config C2 --frame-kind $kind
run X=tightcoh --config C2 --frame-kind $kind
done; done
for cfg in C1 C3 C5; do
echo "== $cfg natural"
for rep in 1 2; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run X=tightcoh --config $cfg
done; done
} 2>&1 | tee $O/ab.log
python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('frame_kinds')))" | tee $O/kinds.json
