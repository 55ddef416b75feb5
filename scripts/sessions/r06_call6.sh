#!/bin/bash
# round 6, session 6: symmetric filter stage, second run of the steps that hold pixels of non-palindromic bank rows.
# Parity slice first; then three-way A/B on one box: eight-load stage (the previous library's choice for filterbin_2_10: 50 rows > 16) /
# symmetric stage + OLD redo forced on it (prev, RAISR_HIP_SYM_MAX_ROWS=64) / symmetric stage + second run (new, RAISR_HIP_SYM_MAX_ROWS=64).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call6; mkdir -p $O
D=$PWD/video-super-resolution-library_amd
timeout 1200 python -m pytest tests/test_gpu_sym_mixed.py tests/test_gpu_parity.py -q -x -m gpu -k "not fuzz" 2>&1 | tail -4 | tee $O/tests.log
RAISR_HIP_SYM_MAX_ROWS=64 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_photos.py -q -x -m gpu -k "10b or C5 or C2b or photo" 2>&1 | tail -4 | tee $O/tests_maxrows64.log
run() {  # label, lib ("" = in-tree), max_rows ("" = default), bench args
  echo -n "$1: "
  if [ -n "$2" ]; then export RAISR_HIP_LIB=$2; else unset RAISR_HIP_LIB; fi
  if [ -n "$3" ]; then export RAISR_HIP_SYM_MAX_ROWS=$3; else unset RAISR_HIP_SYM_MAX_ROWS; fi
  python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 "${@:4}" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], {k: round(v,4) for k,v in d['kernels_isolated_ms'].items()})"
  unset RAISR_HIP_LIB RAISR_HIP_SYM_MAX_ROWS
}
for args in "--config C5 --frames-per-step 96" "--config C2b --frames-per-step 384" "--config C2b --frames-per-step 384 --frame-kind photo" "--config C2"; do
  echo "== $args" | tee -a $O/ab.log
  for r in 1 2 3; do
    run "new  second-run (max rows 64)" "" 64 $args 2>&1 | tee -a $O/ab.log
    run "prev eight-load (default)    " "$D/_exp/libraisr_prev.so" "" $args 2>&1 | tee -a $O/ab.log
    run "prev old redo  (max rows 64) " "$D/_exp/libraisr_prev.so" 64 $args 2>&1 | tee -a $O/ab.log
  done
done
