#!/bin/bash
# GPU session 6 of round 3: pin cache with exact ranges, wave priority in persistent / non-persistent grids, full suite, bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call6; mkdir -p $O
RAISR_HIP_PIN_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_host_api.py -m gpu -x -q -k "registered_planes or async_submit" > $O/pin_test.log 2>&1; tail -3 $O/pin_test.log
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
B="python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
cp $D/libraisr_hip.so /tmp/base.so
cp $D/_exp/libraisr_persistprio.so $D/libraisr_hip.so
echo "== persistent static 4 + setprio, lanes 1"; RAISR_HIP_PERSIST=4 $B --lanes 1 2>/dev/null | show
echo "== persistent static 4 + setprio, lanes 4"; RAISR_HIP_PERSIST=4 $B 2>/dev/null | show
cp $D/_exp/libraisr_prio.so $D/libraisr_hip.so
echo "== non-persistent + setprio, lanes 1"; $B --lanes 1 2>/dev/null | show
echo "== non-persistent + setprio, lanes 4"; $B 2>/dev/null | show
cp /tmp/base.so $D/libraisr_hip.so
echo "== base, lanes 1"; $B --lanes 1 2>/dev/null | show
echo "== base, lanes 4"; $B 2>/dev/null | show
RAISR_HIP_BANDS=1 python scripts/e2e_probe.py 2>&1 | grep "pinned="
RAISR_HIP_BANDS=2 RAISR_HIP_BAND_CHAIN=0 python scripts/e2e_probe.py 2>&1 | grep "pinned="
} > $O/ab.txt 2>&1
cat $O/ab.txt
( time python bench.py ) > $O/bench_full.json 2> $O/bench_full.err; tail -c 3000 $O/bench_full.json | head -c 3000; tail -5 $O/bench_full.err
