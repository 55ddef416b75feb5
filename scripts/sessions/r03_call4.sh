#!/bin/bash
# GPU session 4 of round 3: full suite, C4 after the binary16 changes, plugin path (registered planes, bands, async), persistent variants, certify campaign
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call4; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
B="python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
echo "== C4"; $B --config C4 2>/dev/null | show
echo "== C4 lanes 2"; $B --config C4 --lanes 2 2>/dev/null | show
echo "== C4 lanes 8"; $B --config C4 --lanes 8 2>/dev/null | show
echo "== C2"; $B 2>/dev/null | show
for b in 1 2 3 4 6; do RAISR_HIP_BANDS=$b python scripts/e2e_probe.py 2>&1 | grep "pinned="; done
RAISR_HIP_PIN=0 python scripts/e2e_probe.py 2>&1 | grep "pinned="
python scripts/async_probe.py 2>&1 | grep "async depth"
cp $D/libraisr_hip.so /tmp/base.so
for v in persist persist3 persistlb; do
  cp $D/_exp/libraisr_$v.so $D/libraisr_hip.so
  n=4; [ $v = persist3 ] && n=3
  echo "== $v, $n workgroups per CU, lanes 1"; RAISR_HIP_PERSIST=$n $B --lanes 1 2>/dev/null | show
done
cp /tmp/base.so $D/libraisr_hip.so
echo "== non-persistent, lanes 1"; $B --lanes 1 2>/dev/null | show
} > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 1200 python scripts/certify_campaign.py 16 > $O/certify.log 2>&1; tail -12 $O/certify.log
cp gpurun_out/certify_campaign.json $O/ 2>/dev/null
