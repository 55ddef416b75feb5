#!/bin/bash
# GPU session 29 of round 3: in-tile worklist of 160 / 256 / 320 entries (LDS allows 336): all configurations and frame kinds
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call29; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
cp $D/libraisr_hip.so /tmp/base.so
for v in base list160 list256 list320 list160 list256 list320; do
  if [ $v = base ]; then cp /tmp/base.so $D/libraisr_hip.so; else cp $D/_exp/libraisr_$v.so $D/libraisr_hip.so; fi
  for k in natural random checker constant; do echo -n "$v C2 $k: "; $B --frame-kind $k 2>/dev/null | show; done
  echo -n "$v C1 natural: "; $B --config C1 2>/dev/null | show
  echo -n "$v C1 random: "; $B --config C1 --frame-kind random 2>/dev/null | show
  echo -n "$v C5: "; $B --config C5 --steps 3 2>/dev/null | show
done
cp /tmp/base.so $D/libraisr_hip.so
} > $O/ab.txt 2>&1
cat $O/ab.txt
