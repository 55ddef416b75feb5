#!/bin/bash
# round 6, final evidence 4 (final sources: two comments differ from evidence 3): the WHOLE GPU suite on the final kernel sources
# (what the driver runs at round end), smoke(), per-configuration rocprofv3 profiles (kernel trace + PMC passes) of C2 and C4, and the
# default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r06_final4; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
for cfg in C2 C4; do
  timeout 600 bash scripts/profile_gpu.sh r06d_$cfg --config $cfg > $O/profile_$cfg.log 2>&1
done
cd $R && timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
