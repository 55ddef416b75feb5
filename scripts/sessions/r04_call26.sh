#!/bin/bash
# round 4, call 26: k_blend rewritten (4 columns x 4 rows per thread, no LDS, word stores) against the LDS-tile version (head2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call26; mkdir -p $O
D=video-super-resolution-library_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_host_api.py tests/test_gpu_host_fuzz.py tests/test_gpu_bands.py tests/test_gpu_stream.py -q -x -m gpu 2>&1 | tail -12 | tee $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1 C5; do
echo "== $cfg"
for rep in 1 2; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_head2.so --config $cfg
run X=tree --config $cfg
done; done
} 2>&1 | tee $O/ab.log
