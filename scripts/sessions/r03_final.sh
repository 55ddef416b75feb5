#!/bin/bash
# Final GPU session of a round: full -m gpu suite, the bench line, rocprofv3 kernel stats + PMC passes for every BASELINE configuration.
# usage (GPU box): scripts/r03_final.sh <round tag, e.g. r03>
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${TAG}_final; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -4 $O/bench.err
for c in C2 C3 C4 C5 C1; do timeout 900 scripts/profile_gpu.sh ${TAG}_$c --config $c > $O/prof_$c.log 2>&1; echo "profiled $c"; done
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
