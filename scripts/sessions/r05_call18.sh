#!/bin/bash
# round 5, call 18: worklist of the certified stage -- table entry requested at the start of the hash stage, exact-tensor weights before
# the table's barrier; parity + certify subset, A/B against the previous library
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call18; mkdir -p $O
D=video-super-resolution-library_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_certify.py -q -x -m gpu 2>&1 | tail -3 | tee $O/tests.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
for cfg in C2 C1 C5; do
echo "== $cfg natural"
for rep in 1 2 3; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run X=wlprefetch --config $cfg
done; done
echo "== C2 random"
for rep in 1 2; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config C2 --frame-kind random
run X=wlprefetch --config C2 --frame-kind random
done
} 2>&1 | tee $O/ab.log
