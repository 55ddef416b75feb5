#!/bin/bash
# round 4, call 15: the exact path's table entry and Gaussian weights requested before the separable passes (two global round trips off the
# worklist's critical path) -- parity, A/B against the previous library, phase shares (development build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call15; mkdir -p $O
D=video-super-resolution-library_amd
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_certify.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -1 $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
for cfg in C2 C1 C5; do
echo "== $cfg"
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run X=new --config $cfg
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_prev.so --config $cfg
run X=new --config $cfg
done
RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so python scripts/phase_cycles.py C2 4 2>&1 | grep -v "^ASM\|^---\|RAISR\|^$\|amdgpu.ids"
} 2>&1 | tee $O/ab.log
