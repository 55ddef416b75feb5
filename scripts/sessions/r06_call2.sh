#!/bin/bash
# round 6, session 2: where does real picture content cost time?  Per-photo fps + worklist tile statistics (C2, C1, C2b).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call2; mkdir -p $O
timeout 600 python scripts/photo_kinds_probe.py C2 24 192 2>&1 | grep -v amdgpu.ids | tee $O/photo_kinds_C2.log
timeout 600 python scripts/photo_kinds_probe.py C1 12 768 2>&1 | grep -v amdgpu.ids | tee $O/photo_kinds_C1.log
timeout 600 python scripts/photo_kinds_probe.py C2b 12 192 2>&1 | grep -v amdgpu.ids | tee $O/photo_kinds_C2b.log
cp gpurun_out/photo_kinds_*.json $O/
