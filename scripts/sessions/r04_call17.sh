#!/bin/bash
# round 4, call 17: ceiling probes of the filter stage (development builds, output wrong): coefficient loads for every 2nd / 4th / 16th
# step only (what sharing coefficient rows between pixels could gain at most), window values from 8 LDS reads per row instead of 128,
# both; then a wide fuzz sweep (1500 cases, second seed) over the round-4 kernels.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call17; mkdir -p $O
D=video-super-resolution-library_amd
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1 C5; do
echo "== $cfg"
for rep in 1 2; do
for v in dev reuse2 reuse4 reuse16 nowin nowin_reuse16; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_$v.so --config $cfg
done; done; done
} 2>&1 | tee $O/ab.log
RAISR_FUZZ_N=1500 RAISR_FUZZ_SEED=20260929 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/fuzz.log
