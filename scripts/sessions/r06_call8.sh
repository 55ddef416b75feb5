#!/bin/bash
# round 6, session 8: (a) one workgroup barrier fewer per listed tile (the exact path's table shares the barrier behind the 16-lane
# tensors); (b) symmetric filter stage with a per-row choice of 16 / 32 coefficient bytes per lane for banks with 17..64 non-palindromic
# rows (filterbin_2_10: C5, C2b), thresholds 0..3.  Parity slice first; prev = the previous commit's library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call8; mkdir -p $O
D=$PWD/video-super-resolution-library_amd
timeout 1500 python -m pytest tests/test_gpu_sym_mixed.py tests/test_gpu_parity.py tests/test_gpu_certify.py tests/test_gpu_class1.py -q -x -m gpu -k "not fuzz" 2>&1 | tail -4 | tee $O/tests.log
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_photos.py -q -x -m gpu 2>&1 | tail -4 | tee -a $O/tests.log
run() {  # label, lib ("" = in-tree), "ENV=VAL ..." , bench args
  echo -n "$1: "
  if [ -n "$2" ]; then export RAISR_HIP_LIB=$2; else unset RAISR_HIP_LIB; fi
  env $3 python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 "${@:4}" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], {k: round(v,4) for k,v in d['kernels_isolated_ms'].items()})"
  unset RAISR_HIP_LIB
}
for args in "--config C2" "--config C1" "--config C2 --frame-kind photo"; do
  echo "== $args" | tee -a $O/ab.log
  for r in 1 2 3; do
    run "new " "" "X=1" $args 2>&1 | tee -a $O/ab.log
    run "prev" "$D/_exp/libraisr_prev.so" "X=1" $args 2>&1 | tee -a $O/ab.log
  done
done
for args in "--config C5 --frames-per-step 96" "--config C2b --frames-per-step 384" "--config C2b --frames-per-step 384 --frame-kind photo"; do
  echo "== $args" | tee -a $O/ab.log
  for r in 1 2; do
    run "prev (eight-load stage)          " "$D/_exp/libraisr_prev.so" "X=1" $args 2>&1 | tee -a $O/ab.log
    run "new, per-row choice off           " "" "RAISR_HIP_MIX_MAX_ROWS=0" $args 2>&1 | tee -a $O/ab.log
    for mm in 0 1 2 3; do
      run "new, per-row choice, mix_max = $mm" "" "RAISR_HIP_MIX_MAX=$mm" $args 2>&1 | tee -a $O/ab.log
    done
  done
done
