#!/bin/bash
# round 4, call 11: rocprofv3 kernel-trace stats + PMC passes of every BASELINE configuration on the final kernels (scripts/profile_gpu.sh);
# summarised in the build container by scripts/summarize_profiles.py r04_Cn -> profiles/r04_Cn_*
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in C2 C1 C3 C4 C5; do
  bash scripts/profile_gpu.sh r04_$cfg --config $cfg > gpurun_out/prof_r04_$cfg.log 2>&1
  echo "$cfg: $(tail -1 gpurun_out/prof_r04_$cfg.log)"
  # keep the merge small: raw traces are not needed, the stats / counter csv files are
  find gpurun_out/prof_r04_$cfg -name "*kernel_trace.csv" -delete 2>/dev/null
  find gpurun_out/prof_r04_$cfg -name "*.db" -delete 2>/dev/null
done
du -sh gpurun_out/prof_r04_* | tail -5
