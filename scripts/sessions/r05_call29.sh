#!/bin/bash
# round 5, call 29: do the four waves of a tile want to run in step?  (Moving the hash stage's last barrier into the filter stage let them
# drift and cost 11 %.)  rowil: wave w filters tile rows w, w + 4, w + 8, w + 12 -- the four waves work on ADJACENT rows at any time
# (today: rows 4 apart); rowsync: a workgroup barrier between the rows of the filter stage; rowboth: both.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call29; mkdir -p $O
D=video-super-resolution-library_amd
run() { echo -n "$1 $2 $3 $4 $5: "; env RAISR_HIP_LIB=$R/$D/_exp/libraisr_$1.so python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
echo "== C2 natural"
for rep in 1 2; do
for v in cur rowil rowsync rowboth; do run $v --config C2; done
done
echo "== C5 natural"
for v in cur rowil rowboth; do run $v --config C5; done
} 2>&1 | tee $O/ab.log
RAISR_HIP_LIB=$R/$D/_exp/libraisr_rowil.so timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2 | tee $O/tests_rowil.log
