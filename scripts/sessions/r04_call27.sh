#!/bin/bash
# round 4, call 27: wide fuzz sweeps over the final kernels (device layer: 1200 cases, third seed; plugin host path: 200 cases)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_call27; mkdir -p $O
RAISR_FUZZ_N=1200 RAISR_FUZZ_SEED=20260930 RAISR_FUZZ_MAX_W=200 RAISR_FUZZ_MAX_H=140 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/fuzz.log
RAISR_HOST_FUZZ_N=200 timeout 900 python -m pytest tests/test_gpu_host_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/host_fuzz.log
