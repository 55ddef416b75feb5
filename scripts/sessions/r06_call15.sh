#!/bin/bash
# round 6, session 15: k_hashfilter16's structure tensor -- the first pair rows of the NEXT patch column requested while the current one
# computes (a column's first taps start without waiting for the LDS): parity of the binary16 pipeline on each build, then A/B on C4.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call15; mkdir -p $O
for n in 2 3 5; do
  RAISR_HIP_LIB=$PWD/video-super-resolution-library_amd/_exp/libraisr_t16pre$n.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fp16 and 96x64" 2>&1 | tail -2 | tee -a $O/tests.log
done
for args in "--config C4" "--config C4 --frame-kind photo"; do
  echo "== $args" | tee -a $O/ab.log
  bash scripts/ab_lib_bench.sh "t16pre2 t16pre3 t16pre5" 3 $args 2>&1 | tee -a $O/ab.log
done
