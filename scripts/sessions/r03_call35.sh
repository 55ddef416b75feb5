#!/bin/bash
# GPU session 35 of round 3: kernel stores into page-locked host memory (a copy kernel instead of the copy engine for the row ranges?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call35; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/zero_copy_probe.hip -o /tmp/zcp 2>/dev/null && timeout 120 /tmp/zcp > $O/zcp.txt 2>&1
cat $O/zcp.txt
sed 's/copy_k<<<256, 256/copy_k<<<32, 256/g' scripts/zero_copy_probe.hip > /tmp/zcp32.hip; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 /tmp/zcp32.hip -o /tmp/zcp32 2>/dev/null && timeout 120 /tmp/zcp32 > $O/zcp32.txt 2>&1
echo "== 32 workgroups"; cat $O/zcp32.txt
