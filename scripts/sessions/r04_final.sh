#!/bin/bash
# round 4, final session: the driver's GPU suite command, the bench line, smoke -- for the record of the committed state
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_final; mkdir -p $O
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.log 2>&1; grep -a "passed\|failed" $O/gpu_suite.log | tail -1
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -4 $O/bench.err
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
