#!/bin/bash
# GPU session 14 of round 3: plugin path without the default pin cache (page-locked planes come from RNLHandler_HostAlloc):
# host-API tests, fps of the synchronous and the asynchronous entry per kind of plane memory, bench, then the driver's suite command repeated
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_stream.py -m gpu -x -q > $O/hostapi.log 2>&1; tail -3 $O/hostapi.log
{
HOSTALLOC=1 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
HOSTALLOC=0 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
HOSTALLOC=0 RAISR_HIP_PIN=1 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
PIN=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
PIN=1 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
} > $O/probes.txt 2>&1; cat $O/probes.txt
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -3 $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['end_to_end'], indent=1)); print(d['roofline']['traffic'], d['value'])"
gcc -shared -fPIC -o /tmp/abort_trace.so scripts/abort_trace.c
for i in 1 2 3; do
  ( LD_PRELOAD=/tmp/abort_trace.so timeout 600 python -m pytest tests/ -x -q -m gpu ) > $O/suite_$i.log 2>&1
  rc=$?; echo "run $i rc=$rc"; tail -2 $O/suite_$i.log
  if [ $rc -ne 0 ]; then tail -80 $O/suite_$i.log; break; fi
done
