#!/bin/bash
# GPU session 2 of round 3: L1 load-width probe, full GPU suite on the 4-workgroups-per-CU default, persistent-grid experiment with counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
O=gpurun_out/r03_call2; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 scripts/l1_width_probe.hip -o /tmp/l1p 2>/dev/null && /tmp/l1p > $O/l1_width.txt 2>&1
cat $O/l1_width.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
cp $D/libraisr_hip.so /tmp/base.so
B="python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{
echo "== base C2"; $B 2>/dev/null | show
cp $D/_exp/libraisr_persist.so $D/libraisr_hip.so
echo "== persist lib, not persistent"; $B 2>/dev/null | show
for n in 4 3 2 8; do echo "== persistent, $n workgroups per CU"; RAISR_HIP_PERSIST=$n $B 2>/dev/null | show; done
echo "== persistent parity"; RAISR_HIP_PERSIST=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_y_bit_exact and avx512" 2>&1 | tail -2
} > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P="python $R/bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 2 --warmup 1 --lanes 1 --frames-per-step 8"
for v in persist base; do
  [ $v = base ] && cp /tmp/base.so $R/$D/libraisr_hip.so
  [ $v = persist ] && export RAISR_HIP_PERSIST=4 || unset RAISR_HIP_PERSIST
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS -d $R/$O/pmc_${v}_sq -- $P > $R/$O/pmc_${v}_sq.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_FLAT -d $R/$O/pmc_${v}_lds -- $P > $R/$O/pmc_${v}_lds.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $R/$O/pmc_${v}_l1 -- $P > $R/$O/pmc_${v}_l1.log 2>&1
done
cp /tmp/base.so $R/$D/libraisr_hip.so
python $R/scripts/pmc_summarize.py $R/$O/pmc_persist_sq $R/$O/pmc_persist_lds $R/$O/pmc_persist_l1 > $R/$O/pmc_persist.txt 2>&1
python $R/scripts/pmc_summarize.py $R/$O/pmc_base_sq $R/$O/pmc_base_lds $R/$O/pmc_base_l1 > $R/$O/pmc_base.txt 2>&1
echo done
