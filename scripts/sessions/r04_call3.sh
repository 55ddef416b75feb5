#!/bin/bash
# round 4, call 3: one summation tree per row (16 steps merged level by level) -- parity, then A/B on one box against the previous
# commit's library (_exp/libraisr_prev.so: symmetric stage, per-group trees), the ds_bpermute exchange, and the parts of the fused kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call3; mkdir -p $O
D=video-super-resolution-library_amd
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -2 $O/parity.log
( RAISR_HIP_LIB=$R/$D/_exp/libraisr_symbperm.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "96x64 or 134" ) > $O/parity_bperm.log 2>&1; tail -1 $O/parity_bperm.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1 C5; do
echo "== $cfg"
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run RAISR_HIP_SYM=1 --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_symbperm.so --config $cfg
run RAISR_HIP_SYM=0 --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run RAISR_HIP_SYM=1 --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_symbperm.so --config $cfg
done
echo "== parts (dev build): AC_PART 1 = hash stage only, 2 = filter stage only (PATTERN 0: every bank row, 1: one row, 2: sixteen rows)"
for sym in 0 1; do for part in 0 1 2; do
  run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_SYM=$sym RAISR_HIP_AC_PART=$part" --steps 4
done; done
for sym in 0 1; do for pat in 1 2; do
  run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_SYM=$sym RAISR_HIP_AC_PART=2 RAISR_HIP_AC_PATTERN=$pat" --steps 4
done; done
} 2>&1 | tee $O/ab.log
