#!/bin/bash
# round 6, session 10: census blend with four columns per lane (k_blend4 / k_blend4_16: no LDS, no barrier, aligned 4-sample loads,
# neighbours through DPP wave shifts).  Parity of both kernel families at sizes around the wave seams, the parity slice of the suite on
# the new default, then A/B over RAISR_HIP_BLEND_ROWS = 0 (the 64 x 16 LDS-tile kernels) / 4 / 8 / 16 on C2, C4, C1, interleaved.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blend4.py -q -x -m gpu 2>&1 | tail -15 | tee $O/tests.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_host_api.py -q -x -m gpu -k "not fuzz" 2>&1 | tail -6 | tee -a $O/tests.log
run() {  # rows, bench args
  echo -n "rows=$1: "
  RAISR_HIP_BLEND_ROWS=$1 python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 "${@:2}" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], {k: round(v,4) for k,v in d['kernels_isolated_ms'].items()}, {k: round(v,4) for k,v in d['kernels_avg_ms'].items()})"
}
for cfg in C2 C4 C1; do
  echo "== $cfg" | tee -a $O/ab.log
  for r in 1 2 3; do
    for rows in 0 4 8 16; do run $rows --config $cfg 2>&1 | tee -a $O/ab.log; done
  done
done
