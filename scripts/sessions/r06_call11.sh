#!/bin/bash
# round 6, session 11: k_resize3x2 with a 12 x 3 output block per thread (the 1.5x configurations' cheap upscale: 38 -> ~8 vector and
# 3.7 -> 1.3 memory instructions per output pixel).  Direct test against the oracle's upscale, the 1.5x parity cases, C4 at full size,
# then A/B on C4 against the library of commit 31abdad (prev).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blend4.py -q -x -m gpu 2>&1 | tail -15 | tee $O/tests.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py tests/test_gpu_baseline_configs.py -q -x -m gpu -k "1.5x or 1_5 or C4 or geometries or chroma or layouts" 2>&1 | tail -6 | tee -a $O/tests.log
for args in "--config C4" "--config C4 --frame-kind photo"; do
  echo "== $args" | tee -a $O/ab.log
  bash scripts/ab_lib_bench.sh "prev" 3 $args 2>&1 | tee -a $O/ab.log
done
