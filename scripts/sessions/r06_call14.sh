#!/bin/bash
# round 6, session 14: worklist of 256 entries instead of 160 (fits: tensors behind the table, second window copy 1.5 KB further into sV's
# space, 40 912 B of LDS) -- round 3 found "beyond 160 nothing changes" on synthetic frames; photographs overflow 3 % of their tiles.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call14; mkdir -p $O
RAISR_HIP_LIB=$PWD/video-super-resolution-library_amd/_exp/libraisr_list256.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_photos.py -q -x -m gpu -k "not fuzz and (96x64 or photo)" 2>&1 | tail -3 | tee $O/tests.log
for args in "--config C2 --frame-kind photo" "--config C2" "--config C2 --frame-kind random" "--config C1 --frame-kind photo" "--config C1 --frame-kind random" "--config C2b --frame-kind photo"; do
  echo "== $args" | tee -a $O/ab.log
  bash scripts/ab_lib_bench.sh "list256" 3 $args 2>&1 | tee -a $O/ab.log
done
