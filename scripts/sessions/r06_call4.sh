#!/bin/bash
# round 6, session 4: what binds the pipeline -- clock cap vs power cap (scripts/power_cap_probe.sh), share of a wave's life parked at the
# workgroup barriers (development build, timed barriers); the multi-device tests after the ADVICE changes; the default bench with its new legs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stream_multi.py tests/test_gpu_class1.py tests/test_gpu_rccl_broadcast.py -q -x -m gpu 2>&1 | tail -5 | tee $O/tests.log
timeout 600 bash scripts/power_cap_probe.sh > $O/power_cap.out 2>&1; cp gpurun_out/power_cap/log.txt $O/power_cap_log.txt; tail -20 $O/power_cap.out
for cfg in C2 C1 C5; do
  RAISR_HIP_LIB=$PWD/video-super-resolution-library_amd/_exp/libraisr_dev.so timeout 300 python scripts/phase_cycles.py $cfg 4 2>&1 | grep -v amdgpu.ids | tee $O/phase_cycles_$cfg.txt
done
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
