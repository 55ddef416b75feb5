#!/bin/bash
# round 6, session 17: a share of the tiles of a symmetric bank on the eight-load filter stage (9 % fewer vector instructions, twice the
# coefficient bytes): do the CU's four workgroups load the vector ALU and the vector L1 more evenly when they differ?  Parity on one mix,
# then C2 (and C3, whose first pass has the symmetric bank) by eighths of the tiles.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call17; mkdir -p $O
L=$PWD/video-super-resolution-library_amd/_exp/libraisr_symmix.so
RAISR_HIP_SYM_MIX=3 RAISR_HIP_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "not fuzz and not fp16 and 96x64" 2>&1 | tail -2 | tee $O/tests.log
run() {  # label, lib, mix, args
  echo -n "$1 mix=$3: "
  if [ -n "$2" ]; then export RAISR_HIP_LIB=$2; else unset RAISR_HIP_LIB; fi
  RAISR_HIP_SYM_MIX=$3 python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 "${@:4}" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], {k: round(v,4) for k,v in d['kernels_isolated_ms'].items()})"
  unset RAISR_HIP_LIB
}
for cfg in C2 C3; do
  echo "== $cfg" | tee -a $O/ab.log
  for r in 1 2 3; do
    run "in-tree" "" 0 --config $cfg 2>&1 | tee -a $O/ab.log
    for m in 0 1 2 3 4 6 8; do run "symmix " $L $m --config $cfg 2>&1 | tee -a $O/ab.log; done
  done
done
