#!/bin/bash
# round 5, call 22: filter stage with the pair-column pixel order (group g of a step pair filters columns 2g, 2g+1: one 8-byte window
# read per tap and pair).  gfx950 serves a 4-mod-8 ds_read_b64 correctly but at 64 cycles (call 21), so a real version needs two copies
# of the window; this call measures the CEILING first: a timing probe with every read forced to an aligned address (output wrong),
# against the library as committed and against the correct-but-misaligned variant.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call22; mkdir -p $O
D=video-super-resolution-library_amd
RAISR_HIP_LIB=$R/$D/_exp/libraisr_paircol.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sym_mixed.py -q -x -m gpu 2>&1 | tail -3 | tee $O/tests_paircol.log
run() { echo -n "$1 $2 $3: "; env RAISR_HIP_LIB=$R/$D/_exp/libraisr_$1.so python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d.get('kernels_isolated_ms'))"; }
{
for cfg in C2 C5 C1; do
echo "== $cfg natural"
for rep in 1 2 3; do
run prev --config $cfg
run paircol_alignprobe --config $cfg
done
run paircol --config $cfg
done
} 2>&1 | tee $O/ab.log
