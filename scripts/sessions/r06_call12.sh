#!/bin/bash
# round 6, session 12: does the census blend fit beside the main kernel's waves?  k_hashfilter_ac takes 4 x 104 (symmetric stage) or
# 4 x 112 (eight-load stage) of a SIMD's 512 registers: 96 / 64 are left for a side kernel's wave.  Lean variants of k_blend4 (rows
# requested one / two ahead instead of six: 56-65 registers) against the default (8 rows per wave, 94 / 105 registers), interleaved.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call12; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_blend4.py -q -x -m gpu 2>&1 | tail -5 | tee $O/tests.log
for rows in 41 81 42; do
  RAISR_HIP_BLEND_ROWS=$rows timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "test_y_bit_exact and 96x64" 2>&1 | tail -2 | tee -a $O/tests.log
done
run() {  # rows, bench args
  echo -n "rows=$1: "
  RAISR_HIP_BLEND_ROWS=$1 python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 "${@:2}" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], {k: round(v,4) for k,v in d['kernels_isolated_ms'].items()}, {k: round(v,4) for k,v in d['kernels_avg_ms'].items()})"
}
for cfg in C1 C5 C2 C3 C4; do
  echo "== $cfg" | tee -a $O/ab.log
  for r in 1 2 3; do
    for rows in 8 4 41 81 42; do run $rows --config $cfg 2>&1 | tee -a $O/ab.log; done
  done
done
