#!/bin/bash
# round 4, call 9: fused blend epilogue (interior pixels blended inside k_hashfilter_ac, k_blend_edges for the rest) -- parity, then A/B on
# one box; the synchronous plugin path with pageable planes again (copy threads x streaming stores), standalone
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call9; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_batch.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_bands.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -3 $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C3 C5 C1; do
echo "== $cfg"
run RAISR_HIP_FUSED_BLEND=0 --config $cfg
run RAISR_HIP_FUSED_BLEND=1 --config $cfg
run RAISR_HIP_FUSED_BLEND=0 --config $cfg
run RAISR_HIP_FUSED_BLEND=1 --config $cfg
done
echo "== synchronous RNLHandler_Process, 1080p->4K yuv420p (scripts/e2e_probe.py)"
for th in 2 4 8; do for nt in 0 1; do
  echo -n "pageable threads=$th nt=$nt: "; RAISR_HIP_COPY_THREADS=$th RAISR_HIP_COPY_NT=$nt N=300 python scripts/e2e_probe.py 2>&1 | grep fps
done; done
echo -n "page-locked (HostAlloc): "; HOSTALLOC=1 N=300 python scripts/e2e_probe.py 2>&1 | grep fps
} 2>&1 | tee $O/ab.log
