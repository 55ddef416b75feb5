#!/bin/bash
# round 5, call 14: k_resize2x with an 8 x 2 output block per thread -- whole GPU suite (resize paths: luma, chroma, strided, shifted,
# 16-bit, odd sizes), A/B against the previous library
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call14; mkdir -p $O
D=video-super-resolution-library_amd
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee $O/tests.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1 C3 C5; do
echo "== $cfg natural"
for rep in 1 2 3; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run X=resize8x2 --config $cfg
done; done
} 2>&1 | tee $O/ab.log
