#!/bin/bash
# GPU session 16 of round 3: pageable planes through the library's own bounce memory (csrc/host_copy.h): tests, rates of the
# synchronous and the asynchronous plugin entry per kind of plane memory / row ranges / copy threads, then the driver's suite command repeated
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_stream.py tests/test_gpu_bands.py -m gpu -x -q > $O/hostapi.log 2>&1; tail -3 $O/hostapi.log
{
for th in 1 2 4 8; do for ch in 1 3; do
  echo -n "threads=$th chunks=$ch "; RAISR_HIP_COPY_THREADS=$th RAISR_HIP_CHUNKS=$ch PIN=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
done; done
echo -n "runtime path (RAISR_HIP_BOUNCE=0) chunks=1 "; RAISR_HIP_BOUNCE=0 RAISR_HIP_CHUNKS=1 PIN=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
echo -n "registered planes "; PIN=1 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
HOSTALLOC=1 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
HOSTALLOC=0 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
HOSTALLOC=0 RAISR_HIP_COPY_THREADS=8 timeout 300 python scripts/async_probe.py 2>&1 | grep "async depth"
} > $O/probes.txt 2>&1; cat $O/probes.txt
gcc -shared -fPIC -o /tmp/abort_trace.so scripts/abort_trace.c
for i in 1 2 3 4 5; do
  ( RAISR_TEST_TRACE_PTRS=1 LD_PRELOAD=/tmp/abort_trace.so timeout 600 python -m pytest tests/ -x -q -m gpu ) > $O/suite_$i.log 2>&1
  rc=$?; echo "run $i rc=$rc"; tail -2 $O/suite_$i.log
  if [ $rc -ne 0 ]; then grep -a -A40 "abort_trace\] tail of fd 2" $O/suite_$i.log | tail -60; break; fi
done
