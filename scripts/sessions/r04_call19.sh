#!/bin/bash
# round 4, call 19: hand-pipelined symmetric filter loop (coefficient loads K step pairs ahead, window reads one pair ahead,
# sched_barrier per pair) against the compiler's own schedule
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call19; mkdir -p $O
D=video-super-resolution-library_amd
RAISR_HIP_LIB=$R/$D/_exp/libraisr_pipe3.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2 | tee $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C3; do
echo "== $cfg"
for rep in 1 2; do
run X=tree --config $cfg
for K in 1 2 3; do run RAISR_HIP_LIB=$R/$D/_exp/libraisr_pipe$K.so --config $cfg; done
done; done
} 2>&1 | tee $O/ab.log
