#!/bin/bash
# round 5, call 5: the synchronous plugin path with pageable and page-locked planes, luma uploaded in row ranges or at once
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call5; mkdir -p $O
{
for rep in 1 2 3; do
for up in 0 1; do
for mode in "PIN=0 HOSTALLOC=0" "PIN=0 HOSTALLOC=1"; do
echo -n "UPCHUNKS=$up $mode: "; env RAISR_HIP_UPCHUNKS=$up $mode N=800 python scripts/e2e_probe.py 2>&1 | grep fps
done; done; done
echo "== 8 copy threads"
for up in 0 1; do echo -n "UPCHUNKS=$up threads=8: "; env RAISR_HIP_UPCHUNKS=$up RAISR_HIP_COPY_THREADS=8 N=800 python scripts/e2e_probe.py 2>&1 | grep fps; done
echo "== chunks 2 / 4 with range uploads"
for ch in 2 4 5; do echo -n "CHUNKS=$ch: "; env RAISR_HIP_CHUNKS=$ch N=800 python scripts/e2e_probe.py 2>&1 | grep fps; echo -n "CHUNKS=$ch page-locked: "; env RAISR_HIP_CHUNKS=$ch HOSTALLOC=1 N=800 python scripts/e2e_probe.py 2>&1 | grep fps; done
} 2>&1 | tee $O/e2e.log
