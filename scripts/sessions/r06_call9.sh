#!/bin/bash
# round 6, session 9: binary16 hash stage, gradient tile built by walking down the columns (3 LDS reads per row instead of 8 per entry):
# parity of the binary16 pipeline, then A/B on C4 against the library of commit f4c3747 (prev); parity slice of the fp32 kernels on the
# final tree of this series (second runs with the shifted load, one barrier fewer per listed tile).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call9; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fold16.py tests/test_gpu_batch.py tests/test_gpu_sym_mixed.py -q -x -m gpu -k "not fuzz" 2>&1 | tail -4 | tee $O/tests.log
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_pipelines.py -q -x -m gpu 2>&1 | tail -4 | tee -a $O/tests.log
for args in "--config C4" "--config C4 --frame-kind photo"; do
  echo "== $args" | tee -a $O/ab.log
  bash scripts/ab_lib_bench.sh "prev" 4 $args 2>&1 | tee -a $O/ab.log
done
export RAISR_HIP_LIB=$PWD/video-super-resolution-library_amd/_exp/libraisr_dev.so
python scripts/phase_cycles.py C4 4 2>/dev/null | tee $O/phase_cycles_C4.txt
