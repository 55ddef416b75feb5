#!/bin/bash
# GPU session 25 of round 3: experiment -- binary16 gradient tile for 8-bit samples (32.7 KB LDS) + 96 VGPRs (64 B scratch): 5 workgroups per CU
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call25; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
cp $D/libraisr_hip.so /tmp/base.so
for v in base occ5 base occ5; do
  if [ $v = base ]; then cp /tmp/base.so $D/libraisr_hip.so; else cp $D/_exp/libraisr_$v.so $D/libraisr_hip.so; fi
  for k in natural random; do echo -n "$v C2 $k: "; $B --frame-kind $k 2>/dev/null | show; done
  echo -n "$v C2 lanes 1: "; $B --lanes 1 2>/dev/null | show
  echo -n "$v C1: "; $B --config C1 2>/dev/null | show
  echo -n "$v C3: "; $B --config C3 2>/dev/null | show
done
cp $D/_exp/libraisr_occ5.so $D/libraisr_hip.so
echo "== occ5 parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -2
cp /tmp/base.so $D/libraisr_hip.so
} > $O/ab.txt 2>&1
cat $O/ab.txt
