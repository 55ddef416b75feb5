#!/bin/bash
# round 4, call 21: five workgroups per CU once more, on the hand-pipelined symmetric kernel (97 VGPRs by itself): binary16 gradient
# tile for 8-bit samples + 88-entry worklist = 32 768 B of LDS, __launch_bounds__(256, 5)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call21; mkdir -p $O
D=video-super-resolution-library_amd
RAISR_HIP_LIB=$R/$D/_exp/libraisr_occ5b.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2 | tee $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C3 C1; do
echo "== $cfg"
for rep in 1 2; do
run X=tree --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_occ5b.so --config $cfg
done; done
} 2>&1 | tee $O/ab.log
