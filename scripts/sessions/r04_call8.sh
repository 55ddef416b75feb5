#!/bin/bash
# round 4, call 8: P010-style device frames (sample shift), host-path tests after the bounce / SetRes changes; the synchronous plugin path
# with pageable planes: copy threads x streaming stores, standalone vs inside bench.py; ceiling of a fused blend epilogue (dev build);
# hunt for the round-3 page fault (dev build, RAISR_HIP_BOUNCE=0, the failing test's shape in a loop)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call8; mkdir -p $O
D=video-super-resolution-library_amd
( timeout 1200 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_host_fuzz.py tests/test_gpu_stream.py tests/test_gpu_bench_launch.py -x -q -m gpu ) > $O/host.log 2>&1; tail -2 $O/host.log
{
echo "== synchronous RNLHandler_Process, 1080p->4K yuv420p (scripts/e2e_probe.py)"
for th in 2 4 6 8; do for nt in 0 1; do
  echo -n "pageable threads=$th nt=$nt: "; RAISR_HIP_COPY_THREADS=$th RAISR_HIP_COPY_NT=$nt N=300 python scripts/e2e_probe.py 2>&1 | tail -1
done; done
echo -n "page-locked (HostAlloc): "; HOSTALLOC=1 N=300 python scripts/e2e_probe.py 2>&1 | tail -1
echo "== ceiling of a fused blend epilogue: C2 / C3 / C5 with k_blend gone (dev build, output wrong)"
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
for cfg in C2 C3 C5; do
  run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so" --config $cfg
  run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_SKIP_BLEND=1" --config $cfg
done
} 2>&1 | tee $O/ab.log
echo "== page-fault hunt: dev build, RAISR_HIP_BOUNCE=0, test_process_accepts_strided_and_padded_planes-like loop" | tee -a $O/ab.log
for i in 1 2 3 4 5 6; do
  RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_BOUNCE=0 RAISR_HIP_DEV_BUILD=1 timeout 300 python -m pytest tests/test_gpu_host_api.py -q -m gpu -k "strided or padded or fresh" > $O/hunt$i.log 2>&1; echo "run $i: rc=$? $(tail -1 $O/hunt$i.log)" | tee -a $O/ab.log
done
