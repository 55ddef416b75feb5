#!/bin/bash
# round 6, session 16: H pass of the separable tensor on v_pk_fma_f32 (two rows per instruction on the register pairs the row vectors
# already are: 132 -> 66 instructions per lane and tile, no moves added) -- parity, then A/B on C2 / C1 / C5.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call16; mkdir -p $O
RAISR_HIP_LIB=$PWD/video-super-resolution-library_amd/_exp/libraisr_hpk.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_photos.py -q -x -m gpu -k "not fuzz and not fp16 and (96x64 or photo)" 2>&1 | tail -3 | tee $O/tests.log
for args in "--config C2" "--config C2 --frame-kind photo" "--config C1" "--config C5"; do
  echo "== $args" | tee -a $O/ab.log
  bash scripts/ab_lib_bench.sh "hpk" 3 $args 2>&1 | tee -a $O/ab.log
done
