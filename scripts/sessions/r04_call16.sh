#!/bin/bash
# round 4, call 16: the filter stage's row loop unrolled 2x / 4x (does the next row's prologue overlap the current row's tail?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call16; mkdir -p $O
D=video-super-resolution-library_amd
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1; do
echo "== $cfg"
for rep in 1 2; do
run X=base --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_unroll2.so --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_unroll4.so --config $cfg
done; done
} 2>&1 | tee $O/ab.log
