#!/bin/bash
# round 5, call 26: five workgroups per CU on the pair-column kernel (-DRAISR_EXP_OCC5: binary16 gradient tile for 8-bit samples, 88-entry
# worklist kept in the window's pad columns -> 32 704 B of LDS, 96 VGPRs): the fifth workgroup lost 2.3 % in round 4, when the filter
# stage was waiting for the LDS -- does it pay now?
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call26; mkdir -p $O
D=video-super-resolution-library_amd
RAISR_HIP_LIB=$R/$D/_exp/libraisr_occ5.so timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2 | tee $O/tests_occ5.log
run() { echo -n "$1 $2 $3 $4 $5: "; env RAISR_HIP_LIB=$R/$D/_exp/libraisr_$1.so python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
echo "== C2 natural"
for rep in 1 2 3; do
run cur --config C2
run occ5 --config C2
done
echo "== C2 random"
run cur --config C2 --frame-kind random
run occ5 --config C2 --frame-kind random
echo "== C3 natural"
run cur --config C3
run occ5 --config C3
} 2>&1 | tee $O/ab.log
