#!/bin/bash
# round 5, call 25: the whole GPU suite in one run on the final tree (pair-column kernel), then wide fuzz sweeps over it on a new seed
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call25; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 | tee $O/suite.log
RAISR_FUZZ_N=1000 RAISR_FUZZ_SEED=20261006 RAISR_FUZZ_MAX_W=260 RAISR_FUZZ_MAX_H=150 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/fuzz.log
RAISR_HOST_FUZZ_N=200 RAISR_HOST_FUZZ_SEED=20261006 timeout 600 python -m pytest tests/test_gpu_host_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/host_fuzz.log
