#!/bin/bash
# GPU session 28 of round 3: experiment -- longer in-tile worklist (96 / 160 entries instead of 48) before a tile falls back to the all-exact routine
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call28; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
cp $D/libraisr_hip.so /tmp/base.so
for v in base list96 list160 base list96 list160; do
  if [ $v = base ]; then cp /tmp/base.so $D/libraisr_hip.so; else cp $D/_exp/libraisr_$v.so $D/libraisr_hip.so; fi
  for k in natural random; do echo -n "$v C2 $k: "; $B --frame-kind $k 2>/dev/null | show; done
  echo -n "$v C1 natural: "; $B --config C1 2>/dev/null | show
  echo -n "$v C1 random: "; $B --config C1 --frame-kind random 2>/dev/null | show
  echo -n "$v C3: "; $B --config C3 2>/dev/null | show
done
cp $D/_exp/libraisr_list96.so $D/libraisr_hip.so
echo "== list96 parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -2
cp /tmp/base.so $D/libraisr_hip.so
} > $O/ab.txt 2>&1
cat $O/ab.txt
