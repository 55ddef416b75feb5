#!/bin/bash
# GPU session 24 of round 3: experiment -- per-wave worklists in the certified hash stage (a wave resolves the uncertain pixels of its own
# rows: no workgroup barrier between its exact path and its filter stage; weights of the exact tensor from LDS).  A/B on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call24; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
cp $D/libraisr_hip.so /tmp/base.so
for v in base wavelist base wavelist; do
  if [ $v = base ]; then cp /tmp/base.so $D/libraisr_hip.so; else cp $D/_exp/libraisr_$v.so $D/libraisr_hip.so; fi
  for k in natural random checker; do echo -n "$v C2 $k: "; $B --frame-kind $k 2>/dev/null | show; done
  echo -n "$v C1: "; $B --config C1 2>/dev/null | show
  echo -n "$v C5: "; $B --config C5 --steps 3 2>/dev/null | show
done
cp $D/_exp/libraisr_wavelist.so $D/libraisr_hip.so
echo "== wavelist parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py tests/test_gpu_pipelines.py tests/test_gpu_hash_unit.py -m gpu -x -q 2>&1 | tail -2
cp /tmp/base.so $D/libraisr_hip.so
} > $O/ab.txt 2>&1
cat $O/ab.txt
