#!/bin/bash
# round 5, call 21: (1) LDS probe -- does gfx950 serve ds_read_b64 at 4-byte-aligned addresses, and at what rate (scripts/lds_b64_probe.hip);
# (2) wide fuzz sweeps over the FINAL round-5 kernels (device layer: 1500 cases on a new seed; plugin host path: 200 cases on a new seed)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call21; mkdir -p $O
timeout 120 video-super-resolution-library_amd/_exp/lds_b64_probe 2>&1 | tee $O/lds_b64_probe.log
RAISR_FUZZ_N=1500 RAISR_FUZZ_SEED=20261005 RAISR_FUZZ_MAX_W=200 RAISR_FUZZ_MAX_H=140 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/fuzz.log
RAISR_HOST_FUZZ_N=200 RAISR_HOST_FUZZ_SEED=20261005 timeout 900 python -m pytest tests/test_gpu_host_fuzz.py -q -x -m gpu 2>&1 | tail -3 | tee $O/host_fuzz.log
