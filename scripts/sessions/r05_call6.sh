#!/bin/bash
# round 5, call 6: pageable planes unpacked by a helper thread (RAISR_HIP_ASYNC_UNPACK) and luma ranges uploaded on the compute stream
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_host_fuzz.py tests/test_gpu_stream.py tests/test_gpu_stream_multi.py tests/test_gpu_bands.py -q -x -m gpu 2>&1 | tail -5 | tee $O/tests.log
{
for rep in 1 2 3; do
for au in 0 1; do for up in 0 1; do
echo -n "ASYNC_UNPACK=$au UPCHUNKS=$up pageable: "; env RAISR_HIP_ASYNC_UNPACK=$au RAISR_HIP_UPCHUNKS=$up N=800 python scripts/e2e_probe.py 2>&1 | grep fps
done; done
for up in 0 1; do echo -n "UPCHUNKS=$up page-locked: "; env RAISR_HIP_UPCHUNKS=$up HOSTALLOC=1 N=800 python scripts/e2e_probe.py 2>&1 | grep fps; done
done
echo "== copy threads"
for th in 2 8; do echo -n "threads=$th: "; env RAISR_HIP_COPY_THREADS=$th N=800 python scripts/e2e_probe.py 2>&1 | grep fps; done
} 2>&1 | tee $O/e2e.log
