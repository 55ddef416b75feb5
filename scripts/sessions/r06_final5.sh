#!/bin/bash
# round 6, final evidence 5 (final sources: k_resize3x2 stores its 12-sample blocks through a 4-byte-aligned vector type; everything else
# as in evidence 4): the WHOLE GPU suite (what the driver runs at round end), smoke(), rocprofv3 profiles (kernel trace + PMC passes) of all
# six configurations, and the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r06_final5; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
for cfg in C2 C4 C1 C2b C3 C5; do
  timeout 700 bash scripts/profile_gpu.sh r06e_$cfg --config $cfg > $O/profile_$cfg.log 2>&1
done
cd $R && timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
