#!/bin/bash
# round 5, call 2: k_fix_ac v2 (one wave per four tiles, 13-load tensors, pipelined), partner-block symmetric stage, multi-device ring:
# new tests + a parity subset, then A/B: round-4 library / in-tile worklist / deferred, and the partner block on C5 (50 rows) and C3
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call2; mkdir -p $O
D=video-super-resolution-library_amd
timeout 1200 python -m pytest tests/test_gpu_sym_mixed.py tests/test_gpu_stream_multi.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_baseline_configs.py tests/test_gpu_host_api.py -q -x -m gpu 2>&1 | tail -15 | tee $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1; do
echo "== $cfg"
for rep in 1 2; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_r04.so --config $cfg
run RAISR_HIP_DEFER=0 --config $cfg
run RAISR_HIP_DEFER=1 --config $cfg
done; done
for cfg in C5 C3; do
echo "== $cfg"
for rep in 1 2; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_r04.so --config $cfg
run "RAISR_HIP_DEFER=0 RAISR_HIP_SYM_MAX_ROWS=16" --config $cfg
run "RAISR_HIP_DEFER=0" --config $cfg
run "RAISR_HIP_DEFER=1" --config $cfg
done; done
} 2>&1 | tee $O/ab.log
