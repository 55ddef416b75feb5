#!/bin/bash
# GPU session 9 of round 3: 64 x 8 tiles (5 / 6 workgroups per CU) vs 64 x 16, NV12-style device frames, timeline of the chunked plugin path
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call9; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_host_api.py -m gpu -x -q -k "nv12 or hipexternal" > $O/nv12.log 2>&1; tail -3 $O/nv12.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
cp $D/libraisr_hip.so /tmp/base.so
for v in tile8 tile8w5; do
  cp $D/_exp/libraisr_$v.so $D/libraisr_hip.so
  echo "== $v parity"; RAISR_HIP_TILE8=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py -m gpu -x -q -k "test_y_bit_exact or certified_buckets_equal" 2>&1 | tail -2
  echo "== $v C2 lanes 1"; RAISR_HIP_TILE8=1 $B --lanes 1 2>/dev/null | show
  echo "== $v C2 lanes 4"; RAISR_HIP_TILE8=1 $B 2>/dev/null | show
  echo "== $v C3"; RAISR_HIP_TILE8=1 $B --config C3 2>/dev/null | show
  echo "== $v C5"; RAISR_HIP_TILE8=1 $B --config C5 2>/dev/null | show
  echo "== $v C1"; RAISR_HIP_TILE8=1 $B --config C1 2>/dev/null | show
done
cp /tmp/base.so $D/libraisr_hip.so
echo "== base C2 lanes 1"; $B --lanes 1 2>/dev/null | show
echo "== base C2 lanes 4"; $B 2>/dev/null | show
echo "== base C1"; $B --config C1 2>/dev/null | show
} > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for ch in 3 1; do
  N=48 RAISR_HIP_CHUNKS=$ch timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_chunks$ch -- python $R/scripts/e2e_probe.py > $R/$O/trace$ch.log 2>&1
done
find $R/$O -name "*trace.csv" | head
echo done
