#!/bin/bash
# round 6, final evidence 4b: rocprofv3 profiles (kernel trace + PMC passes) of the remaining configurations on the final sources
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_final4b; mkdir -p $O
for cfg in C1 C2b C3 C5; do
  timeout 700 bash scripts/profile_gpu.sh r06d_$cfg --config $cfg > $O/profile_$cfg.log 2>&1
done
ls gpurun_out | grep prof_r06d
