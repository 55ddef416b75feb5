#!/bin/bash
# round 5, call 23: pair-column filter stage with the two-copy window (every 8-byte window read aligned and conflict-free):
# parity / certify / sym-mixed / pipelines / bands / fuzz on the candidate (= the tree), then A/B against the library as committed (prev)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call23; mkdir -p $O
D=video-super-resolution-library_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_sym_mixed.py tests/test_gpu_certify.py tests/test_gpu_pipelines.py tests/test_gpu_bands.py tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/tests.log
run() { echo -n "$1 $2 $3 $4 $5: "; env RAISR_HIP_LIB=$R/$D/_exp/libraisr_$1.so python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
for cfg in C2 C5 C1 C3; do
echo "== $cfg natural"
for rep in 1 2 3; do
run prev --config $cfg
run pc --config $cfg
done; done
for kind in constant random checker; do
echo "== C2 $kind"
for rep in 1 2; do
run prev --config C2 --frame-kind $kind
run pc --config C2 --frame-kind $kind
done; done
} 2>&1 | tee $O/ab.log
