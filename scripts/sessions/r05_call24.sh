#!/bin/bash
# round 5, call 24: two small variants of the pair-column filter loop against the tree (pc): the pair's bucket bytes as two 1-byte reads
# instead of one 2-byte read plus unpacking (pc_u8); two pairs of look-ahead now that the loop holds 16 fewer window registers (pc_ahead2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call24; mkdir -p $O
D=video-super-resolution-library_amd
for v in pc_u8 pc_ahead2; do
RAISR_HIP_LIB=$R/$D/_exp/libraisr_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2 | tee $O/tests_$v.log
done
run() { echo -n "$1 $2 $3 $4 $5: "; env RAISR_HIP_LIB=$R/$D/_exp/libraisr_$1.so python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
for cfg in C2 C3 C5; do
echo "== $cfg natural"
for rep in 1 2 3; do
run pc --config $cfg
run pc_u8 --config $cfg
run pc_ahead2 --config $cfg
done; done
} 2>&1 | tee $O/ab.log
