#!/bin/bash
# round 6: wide fuzz sweeps over the final round-6 kernels (k_blend4 / k_blend4_16, 12 x 3 k_resize3x2, 256-entry worklist), HIP vs oracle,
# on seeds no earlier sweep used.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_fuzz; mkdir -p $O
{
echo "# round 6 (scripts/sessions/r06_fuzz.sh): wide fuzz sweeps over the final round-6 kernels, HIP vs oracle"
echo "## device layer: RAISR_FUZZ_N=1500 RAISR_FUZZ_SEED=20260930 RAISR_FUZZ_MAX_W=220 RAISR_FUZZ_MAX_H=140 tests/test_gpu_fuzz.py"
RAISR_FUZZ_N=1500 RAISR_FUZZ_SEED=20260930 RAISR_FUZZ_MAX_W=220 RAISR_FUZZ_MAX_H=140 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 4 2>&1 | tail -4
echo "## plugin host path: RAISR_HOST_FUZZ_N=300 RAISR_HOST_FUZZ_SEED=20260930 tests/test_gpu_host_fuzz.py"
RAISR_HOST_FUZZ_N=300 RAISR_HOST_FUZZ_SEED=20260930 timeout 900 python -m pytest tests/test_gpu_host_fuzz.py -q -m gpu 2>&1 | tail -2
} | tee $O/fuzz.log
