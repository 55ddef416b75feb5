#!/bin/bash
# round 5, call 9: row ranges of the synchronous plugin path (RAISR_HIP_CHUNKS) on the final library, page-locked and pageable planes
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call9; mkdir -p $O
{
for rep in 1 2; do
for ch in 1 2 3 4 5 6 8; do
echo -n "CHUNKS=$ch page-locked: "; env RAISR_HIP_CHUNKS=$ch HOSTALLOC=1 N=800 python scripts/e2e_probe.py 2>&1 | grep fps
echo -n "CHUNKS=$ch pageable: "; env RAISR_HIP_CHUNKS=$ch N=800 python scripts/e2e_probe.py 2>&1 | grep fps
done; done
} 2>&1 | tee $O/chunks.log
