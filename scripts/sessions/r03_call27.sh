#!/bin/bash
# GPU session 27 of round 3: host-path fuzz (new), then the driver's suite command on the relinked library
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_fuzz.py -m gpu -x -q > $O/hostfuzz.log 2>&1; tail -5 $O/hostfuzz.log
( time timeout 1800 python -m pytest tests/ -x -q -m gpu ) > $O/gpu_suite.log 2>&1; grep -a "passed\|failed" $O/gpu_suite.log | tail -1
