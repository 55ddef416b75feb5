#!/bin/bash
# round 5, call 1: deferred exact path (k_hashfilter_ac<DEFER> + k_fix_ac) -- parity suites, then A/B against the in-tile worklist
# (RAISR_HIP_DEFER=0, same library) on the five configurations and the four frame kinds
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipelines.py tests/test_gpu_bands.py tests/test_gpu_certify.py tests/test_gpu_host_api.py -q -x -m gpu 2>&1 | tail -15 | tee $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1 C3 C5; do
echo "== $cfg"
for rep in 1 2; do
run RAISR_HIP_DEFER=0 --config $cfg
run RAISR_HIP_DEFER=1 --config $cfg
done; done
for kind in random constant checker; do
echo "== C2 $kind"
run RAISR_HIP_DEFER=0 --config C2 --frame-kind $kind
run RAISR_HIP_DEFER=1 --config C2 --frame-kind $kind
done
} 2>&1 | tee $O/ab.log
