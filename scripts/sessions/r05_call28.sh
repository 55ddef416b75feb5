#!/bin/bash
# round 5, call 28: the hash stage's last barrier moved into the filter stage (rows without a listed pixel first): the three waves that sat
# at it while one hashes the worklist filter instead.  Parity on the candidate, A/B against the library as committed (cur).
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call28; mkdir -p $O
D=video-super-resolution-library_amd
RAISR_HIP_LIB=$R/$D/_exp/libraisr_late.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sym_mixed.py tests/test_gpu_certify.py tests/test_gpu_bands.py -q -x -m gpu 2>&1 | tail -2 | tee $O/tests_late.log
run() { echo -n "$1 $2 $3 $4 $5: "; env RAISR_HIP_LIB=$R/$D/_exp/libraisr_$1.so python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
echo "== C2 natural"
for rep in 1 2 3; do
run cur --config C2
run late --config C2
done
for cfg in C1 C5; do
echo "== $cfg natural"
for rep in 1 2; do
run cur --config $cfg
run late --config $cfg
done; done
echo "== C2 random"
run cur --config C2 --frame-kind random
run late --config C2 --frame-kind random
} 2>&1 | tee $O/ab.log
