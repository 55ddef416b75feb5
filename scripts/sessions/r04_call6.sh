#!/bin/bash
# round 4, call 6: full GPU suite on the restructured library (frame batches, no fast mode in the product, bounce-memory fences, golden
# fixtures, pinning-kit dry run), then the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_call6; mkdir -p $O
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.log 2>&1; grep -a "passed\|failed\|error" $O/gpu_suite.log | tail -3
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; tail -3 $O/bench.err
