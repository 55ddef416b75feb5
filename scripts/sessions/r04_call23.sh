#!/bin/bash
# round 4, call 23: the eight-load filter loop pipelined by hand like the symmetric one (pipe8_1: 4-byte loads; pipe8_2: two 16-byte
# loads per step from the lane-major bank -- its tail columns are wrong, timing only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call23; mkdir -p $O
D=video-super-resolution-library_amd
RAISR_HIP_LIB=$R/$D/_exp/libraisr_pipe8_1.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2 | tee $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C1 C5 C3; do
echo "== $cfg"
for rep in 1 2; do
run X=tree --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_pipe8_1.so --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_pipe8_2.so --config $cfg
done; done
} 2>&1 | tee $O/ab.log
