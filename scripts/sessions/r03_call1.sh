#!/bin/bash
# GPU session 1 of round 3: A/B of the 4-workgroups-per-CU fused kernel, L1-locality probe of the filter stage, baseline profiles of every config
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call1; mkdir -p $O
cp $D/libraisr_hip.so /tmp/base.so
B="python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
echo "== base C2"; $B 2>/dev/null | show
for pat in 0 1 2; do echo "== base filter-only pattern $pat"; RAISR_HIP_AC_PART=2 RAISR_HIP_AC_PATTERN=$pat $B 2>/dev/null | show; done
echo "== base hash-only"; RAISR_HIP_AC_PART=1 $B 2>/dev/null | show
cp $D/_exp/libraisr_occ4.so $D/libraisr_hip.so
echo "== occ4 parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py -m gpu -x -q 2>&1 | tail -3
echo "== occ4 C2"; $B 2>/dev/null | show
for pat in 0 1 2; do echo "== occ4 filter-only pattern $pat"; RAISR_HIP_AC_PART=2 RAISR_HIP_AC_PATTERN=$pat $B 2>/dev/null | show; done
echo "== occ4 hash-only"; RAISR_HIP_AC_PART=1 $B 2>/dev/null | show
for c in C1 C3 C5; do echo "== occ4 $c"; $B --config $c 2>/dev/null | show; done
echo "== occ4 C2 lanes 1,2,8"; for l in 1 2 8; do $B --lanes $l 2>/dev/null | show; done
cp /tmp/base.so $D/libraisr_hip.so
for c in C1 C3 C4 C5; do echo "== base $c"; $B --config $c 2>/dev/null | show; done
echo "== base C2 again"; $B 2>/dev/null | show
} > $O/ab.txt 2>&1
cat $O/ab.txt
for c in C4 C3 C5 C1; do scripts/profile_gpu.sh r03pre_$c --config $c > $O/prof_$c.log 2>&1; done
echo profiles done
