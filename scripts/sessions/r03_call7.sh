#!/bin/bash
# GPU session 7 of round 3: row-chunked plugin path, CPU baseline scaling on the box's host, C4 counters, suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_host_api.py -m gpu -x -q -k "chunks or row_chunks" > $O/chunk_tests.log 2>&1; tail -3 $O/chunk_tests.log
{
for ch in 1 2 3 4 6 8; do echo -n "chunks=$ch "; RAISR_HIP_CHUNKS=$ch python scripts/e2e_probe.py 2>&1 | grep "pinned="; done
echo -n "chunks=3 pin=0 "; RAISR_HIP_PIN=0 RAISR_HIP_CHUNKS=3 python scripts/e2e_probe.py 2>&1 | grep "pinned="
echo -n "chunks=4 pin=0 "; RAISR_HIP_PIN=0 RAISR_HIP_CHUNKS=4 python scripts/e2e_probe.py 2>&1 | grep "pinned="
python scripts/omp_scale_probe.py 2>&1
} > $O/ab.txt 2>&1
cat $O/ab.txt
scripts/profile_gpu.sh r03mid_C4 --config C4 > $O/prof_C4.log 2>&1
echo done
