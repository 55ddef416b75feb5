#!/bin/bash
# GPU session 5 of round 3: pin-cache failure with logging, band chaining on the plugin path, dynamic persistent grid, C4 counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call5; mkdir -p $O
RAISR_HIP_PIN_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_host_api.py -m gpu -x -q -k "registered_planes or async_submit" > $O/pin_test.log 2>&1; tail -3 $O/pin_test.log
grep "raisr pin" $O/pin_test.log | tail -40 > $O/pin_tail.txt
B="python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
for ch in 1 0; do for b in 1 2 3 4 6 8; do echo -n "chain=$ch "; RAISR_HIP_BAND_CHAIN=$ch RAISR_HIP_BANDS=$b python scripts/e2e_probe.py 2>&1 | grep "pinned="; done; done
cp $D/libraisr_hip.so /tmp/base.so
cp $D/_exp/libraisr_persist.so $D/libraisr_hip.so
for n in 4 8; do echo "== persistent dynamic, $n workgroups per CU, lanes 1"; RAISR_HIP_PERSIST_DYN=$n $B --lanes 1 2>/dev/null | show; done
echo "== persistent dynamic 4, lanes 4"; RAISR_HIP_PERSIST_DYN=4 $B 2>/dev/null | show
echo "== persistent static 4, lanes 1"; RAISR_HIP_PERSIST=4 $B --lanes 1 2>/dev/null | show
echo "== persistent dynamic parity"; RAISR_HIP_PERSIST_DYN=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_y_bit_exact and avx512" 2>&1 | tail -2
cp /tmp/base.so $D/libraisr_hip.so
echo "== non-persistent, lanes 1"; $B --lanes 1 2>/dev/null | show
echo "== C4"; $B --config C4 2>/dev/null | show
} > $O/ab.txt 2>&1
cat $O/ab.txt
scripts/profile_gpu.sh r03mid_C4 --config C4 > $O/prof_C4.log 2>&1
echo done
