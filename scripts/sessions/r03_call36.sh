#!/bin/bash
# GPU session 36 of round 3: experiment -- row ranges (and chroma planes) sent back by a small copy kernel (n workgroups) instead of the copy engine
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call36; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $D/libraisr_hip.so /tmp/base.so; cp $D/_exp/libraisr_d2hk.so $D/libraisr_hip.so
{
for rep in 1 2; do for n in 0 16 32 64; do
  echo -n "copy kernel wgs=$n hostalloc: "; RAISR_HIP_D2H_KERNEL=$n HOSTALLOC=1 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
done; done
echo -n "wgs=32 registered numpy: "; RAISR_HIP_D2H_KERNEL=32 PIN=1 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
echo -n "wgs=32 pageable (bounce): "; RAISR_HIP_D2H_KERNEL=32 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
echo -n "wgs=0 pageable (bounce): "; RAISR_HIP_D2H_KERNEL=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
echo "== async, hostalloc, wgs 0 / 32"
RAISR_HIP_D2H_KERNEL=0 HOSTALLOC=1 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
RAISR_HIP_D2H_KERNEL=32 HOSTALLOC=1 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
echo "== host tests with wgs=32"; RAISR_HIP_D2H_KERNEL=32 timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_host_fuzz.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -2
} > $O/probes.txt 2>&1
cp /tmp/base.so $D/libraisr_hip.so
cat $O/probes.txt
