#!/bin/bash
# round 6, session 5: the exact path's table requested before the separable passes (instead of fetched behind the list's barrier) and the
# 16-lane tensor's weights requested before the window copy: parity slice, then A/B against the previous commit's library on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py tests/test_gpu_class1.py -q -x -m gpu -k "not fuzz" 2>&1 | tail -4 | tee $O/tests.log
for args in "--config C2" "--config C2 --frame-kind photo" "--config C1" "--config C5 --frames-per-step 96" "--config C3 --frames-per-step 384"; do
  echo "== $args" | tee -a $O/ab.log
  bash scripts/ab_lib_bench.sh "prev" 3 $args 2>&1 | tee -a $O/ab.log
done
