#!/bin/bash
# round 4, call 7: rest of the GPU suite after the bench-test fix; experiment: 5 workgroups per CU for the fused kernel (binary16 gradient
# tile, 32 KB of LDS, 96 VGPRs) with the symmetric filter stage -- parity on 8-bit cases, then A/B on C2 / C1
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call7; mkdir -p $O
D=video-super-resolution-library_amd
( RAISR_HIP_LIB=$R/$D/_exp/libraisr_occ5.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_y_bit_exact and (8b_1p_avx512 or 8b_1p_avx2 or 8b_2p_m1)" ) > $O/parity_occ5.log 2>&1; tail -1 $O/parity_occ5.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1; do
echo "== $cfg"
run RAISR_HIP_SYM=1 --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_occ5.so --config $cfg
run RAISR_HIP_SYM=1 --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_occ5.so --config $cfg
done
} 2>&1 | tee $O/ab.log
( time timeout 2400 python -m pytest tests/ -q -m gpu ) > $O/gpu_suite.log 2>&1; grep -a "passed\|failed" $O/gpu_suite.log | tail -3
