#!/bin/bash
# round 5, call 22: filter stage with the pair-column pixel order (one ds_read_b64 per tap and pair of steps instead of one ds_read2_b32):
# parity on the candidate library, then A/B against the library as committed (prev), bucket bytes as one 2-byte read or two 1-byte reads
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call22; mkdir -p $O
D=video-super-resolution-library_amd
for v in paircol paircol_u8; do
RAISR_HIP_LIB=$R/$D/_exp/libraisr_$v.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_sym_mixed.py tests/test_gpu_certify.py -q -x -m gpu 2>&1 | tail -3 | tee $O/tests_$v.log
done
run() { echo -n "$1 $2 $3: "; env RAISR_HIP_LIB=$R/$D/_exp/libraisr_$1.so python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
for cfg in C2 C5 C1 C3; do
echo "== $cfg natural"
for rep in 1 2 3; do
run prev --config $cfg
run paircol --config $cfg
run paircol_u8 --config $cfg
done; done
} 2>&1 | tee $O/ab.log
