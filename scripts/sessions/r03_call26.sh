#!/bin/bash
# GPU session 26 of round 3: row ranges downloaded on two streams in turn (RAISR_HIP_DOWN_LANES=1: one, as before)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call26; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_stream.py tests/test_gpu_bands.py tests/test_gpu_pipelines.py -m gpu -x -q > $O/hostapi.log 2>&1; tail -2 $O/hostapi.log
{
for rep in 1 2; do for l in 1 2; do
  echo -n "lanes=$l registered planes: "; RAISR_HIP_DOWN_LANES=$l PIN=1 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
  echo -n "lanes=$l pageable planes: "; RAISR_HIP_DOWN_LANES=$l PIN=0 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="
done; done
for ch in 2 4; do echo -n "lanes=2 chunks=$ch registered: "; RAISR_HIP_CHUNKS=$ch PIN=1 timeout 300 python scripts/e2e_probe.py 2>&1 | grep "pinned="; done
} > $O/probes.txt 2>&1; cat $O/probes.txt
cd /tmp && export TMPDIR=/tmp
N=48 PIN=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_lanes2 -- python $R/scripts/e2e_probe.py > $R/$O/trace.log 2>&1
echo traced
