#!/bin/bash
# round 6, session 3: the class-1 rule (exactly one-dimensional windows certified by a sign table): its own tests, the certification suites,
# the photo parity tests, then the A/B on one box -- photo_kinds_probe with RAISR_HIP_C1=0 and with the rule on.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_class1.py -q -x -m gpu 2>&1 | tail -8 | tee $O/class1_tests.log
timeout 1200 python -m pytest tests/test_gpu_certify.py tests/test_gpu_photos.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipelines.py -q -x -m gpu 2>&1 | tail -6 | tee $O/suites.log
for c1 in 0 1; do
  RAISR_HIP_C1=$c1 timeout 600 python scripts/photo_kinds_probe.py C2 24 192 2>&1 | grep -v amdgpu.ids > $O/photo_kinds_C2_c1_$c1.log
  RAISR_HIP_C1=$c1 timeout 600 python scripts/photo_kinds_probe.py C1 12 768 2>&1 | grep -v amdgpu.ids > $O/photo_kinds_C1_c1_$c1.log
done
paste -d'|' $O/photo_kinds_C2_c1_0.log $O/photo_kinds_C2_c1_1.log | cut -c1-260
paste -d'|' $O/photo_kinds_C1_c1_0.log $O/photo_kinds_C1_c1_1.log | cut -c1-260
