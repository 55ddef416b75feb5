#!/bin/bash
# round 6, final evidence 1: the WHOLE GPU suite on the final kernel sources (what the driver runs at round end), smoke(), the probe of the
# sustained v_fma_f32 rate, per-configuration rocprofv3 profiles (kernel trace + PMC passes) of C2 and C4.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r06_final1; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/valu_rate_probe.hip -o /tmp/valu_probe 2>/dev/null && /tmp/valu_probe | tee $O/valu_rate_probe.txt
for cfg in C2 C4; do
  timeout 600 bash scripts/profile_gpu.sh r06_$cfg --config $cfg > $O/profile_$cfg.log 2>&1
done
