#!/bin/bash
# GPU session 22 of round 3: row-copy pool with polling workers / byte-count completion: host tests, rates of the pageable path
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_stream.py tests/test_gpu_bands.py -m gpu -x -q > $O/hostapi.log 2>&1; tail -3 $O/hostapi.log
{
for spin in 20000 0; do for th in 2 4 6; do
  echo -n "spin=$spin threads=$th "; RAISR_HIP_COPY_SPIN=$spin RAISR_HIP_COPY_THREADS=$th PIN=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
done; done
echo -n "chunks=4 threads=4 "; RAISR_HIP_CHUNKS=4 PIN=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
echo -n "chunks=2 threads=4 "; RAISR_HIP_CHUNKS=2 PIN=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
echo -n "registered planes "; PIN=1 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
HOSTALLOC=0 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
} > $O/probes.txt 2>&1; cat $O/probes.txt
