#!/bin/bash
# round 4, call 14: what are two of the hash stage's workgroup barriers worth?  TIMING PROBES, output wrong: _exp/libraisr_nobar1.so drops the
# barrier between window staging and the gradient tile, nobar2 the one after the exact path's table staging, nobar12 both
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call14; mkdir -p $O
D=video-super-resolution-library_amd
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
run X=base --config C2
for v in nobar1 nobar2 nobar12; do run RAISR_HIP_LIB=$R/$D/_exp/libraisr_$v.so --config C2; done
run X=base --config C2
for v in nobar1 nobar2 nobar12; do run RAISR_HIP_LIB=$R/$D/_exp/libraisr_$v.so --config C2; done
} 2>&1 | tee $O/ab.log
