#!/bin/bash
# round 5, call 8: k_fix_ac v3 (one lane per entry) -- bit-exactness of the deferred pipeline, then A/B on the test-hooks library
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call8; mkdir -p $O
D=video-super-resolution-library_amd
timeout 1200 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_product_refuses_fast_mode.py tests/test_gpu_bench_launch.py tests/test_gpu_stream_multi.py -q -x -m gpu -k "defer or refuse or hooks or single_process or plugin_api" 2>&1 | tail -6 | tee $O/tests.log
run() { echo -n "$1: "; env RAISR_HIP_LIB=$R/$D/libraisr_hip_testhooks.so $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1 C3 C5; do
echo "== $cfg"
for rep in 1 2; do
run RAISR_HIP_DEFER=0 --config $cfg
run RAISR_HIP_DEFER=1 --config $cfg
done; done
for kind in random checker; do
echo "== C2 $kind"
run RAISR_HIP_DEFER=0 --config C2 --frame-kind $kind
run RAISR_HIP_DEFER=1 --config C2 --frame-kind $kind
done
} 2>&1 | tee $O/ab.log
