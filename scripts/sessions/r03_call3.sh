#!/bin/bash
# GPU session 3 of round 3: full suite (new host path, blend16 / resize rewrites), C4, plugin path with registered planes and bands,
# async plugin path, phase-skewed persistent grid
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call3; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
B="python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
echo "== C4"; $B --config C4 2>/dev/null | show
echo "== C4 lanes 8"; $B --config C4 --lanes 8 2>/dev/null | show
echo "== C2"; $B 2>/dev/null | show
for b in 1 2 3 4 6; do RAISR_HIP_BANDS=$b python scripts/e2e_probe.py 2>&1 | tail -1; done
RAISR_HIP_PIN=0 python scripts/e2e_probe.py 2>&1 | tail -1
python scripts/async_probe.py 2>&1 | tail -4
cp $D/libraisr_hip.so /tmp/base.so
cp $D/_exp/libraisr_persist.so $D/libraisr_hip.so
for sk in 0 300 700 1500; do echo "== persistent 4/CU, skew $sk ticks"; RAISR_HIP_PERSIST=4 RAISR_HIP_PERSIST_SKEW=$sk $B --lanes 1 2>/dev/null | show; done
echo "== non-persistent, lanes 1"; $B --lanes 1 2>/dev/null | show
cp /tmp/base.so $D/libraisr_hip.so
} > $O/ab.txt 2>&1
cat $O/ab.txt
