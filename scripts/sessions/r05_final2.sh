#!/bin/bash
# round 5, last session: the GPU suite on the final tree (the files call 23 did not run -- the device code is the one call 23 tested, only
# two unused experiment switches left the source since), smoke, per-configuration profiles incl. the PMC traffic passes for the final
# kernel sources, the default bench.py run
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_final2; mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu --ignore=tests/test_gpu_parity.py --ignore=tests/test_gpu_baseline_configs.py --ignore=tests/test_gpu_sym_mixed.py --ignore=tests/test_gpu_certify.py --ignore=tests/test_gpu_pipelines.py --ignore=tests/test_gpu_bands.py --ignore=tests/test_gpu_fuzz.py 2>&1 | tail -6 | tee $O/suite_rest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
for cfg in C2 C4 C1 C3 C5; do
  timeout 420 bash scripts/profile_gpu.sh r05_$cfg --config $cfg > $O/profile_$cfg.log 2>&1
done
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
