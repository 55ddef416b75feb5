#!/bin/bash
# round 5, call 4: host path with the luma uploaded in row ranges (RAISR_HIP_UPCHUNKS) -- host-API / fuzz / stream tests, then the
# synchronous plugin path with pageable and page-locked planes, A/B against one upload
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_host_fuzz.py tests/test_gpu_stream.py tests/test_gpu_stream_multi.py tests/test_gpu_bands.py tests/test_gpu_sym_mixed.py -q -x -m gpu 2>&1 | tail -8 | tee $O/tests.log
{
for rep in 1 2 3; do
for up in 0 1; do
for mode in "PIN=0 HOSTALLOC=0" "PIN=0 HOSTALLOC=1"; do
echo -n "UPCHUNKS=$up $mode: "; env RAISR_HIP_UPCHUNKS=$up $mode N=600 python scripts/e2e_probe.py 2>&1 | tail -1
done; done; done
echo "== 8 copy threads"
for up in 0 1; do echo -n "UPCHUNKS=$up threads=8: "; env RAISR_HIP_UPCHUNKS=$up RAISR_HIP_COPY_THREADS=8 N=600 python scripts/e2e_probe.py 2>&1 | tail -1; done
} 2>&1 | tee $O/e2e.log
