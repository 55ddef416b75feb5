#!/bin/bash
# GPU session 18 of round 3: experiment -- bank rows of the frequent (bucket, type) keys cached in the LDS space the hash stage
# leaves behind (56 rows), coefficient reads of fully cached wave steps from LDS instead of the vector L1 (scripts/build_exp.sh hotrows)
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call18; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
B="timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
cp $D/libraisr_hip.so /tmp/base.so
echo "== base natural"; $B 2>/dev/null | show
echo "== base random"; $B --frame-kind random 2>/dev/null | show
cp $D/_exp/libraisr_hotrows.so $D/libraisr_hip.so
export RAISR_HIP_HOT_FILE=$R/scripts/exp_hot_keys_natural.txt
echo "== hotrows parity (natural keys)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -2
for n in 0 16 32 56; do
  echo "== hotrows n=$n natural"; RAISR_HIP_HOT_N=$n $B 2>/dev/null | show
done
echo "== hotrows n=56 natural lanes 1"; $B --lanes 1 2>/dev/null | show
export RAISR_HIP_HOT_FILE=$R/scripts/exp_hot_keys_random.txt
echo "== hotrows n=56 random (random keys)"; $B --frame-kind random 2>/dev/null | show
echo "== hotrows n=56 constant (random keys)"; $B --frame-kind constant 2>/dev/null | show
cp /tmp/base.so $D/libraisr_hip.so
} > $O/ab.txt 2>&1
cat $O/ab.txt
