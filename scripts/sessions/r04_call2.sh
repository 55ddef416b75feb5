#!/bin/bash
# round 4, call 2: symmetric filter stage -- exchange through DPP (product) vs ds_bpermute; parts of the fused kernel (dev build) with
# SYM on / off; texture-path and SQ counters with SYM on / off (--lanes 1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r04_call2; mkdir -p $O
D=video-super-resolution-library_amd
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "96x64 or 134" ) > $O/parity.log 2>&1; tail -2 $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
echo "== exchange: DPP x2 (product) vs ds_bpermute, C2"
run RAISR_HIP_SYM=1
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_symbperm.so
run RAISR_HIP_SYM=1
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_symbperm.so
run RAISR_HIP_SYM=0
echo "== parts (dev build): AC_PART 1 = hash stage only, 2 = filter stage only (PATTERN 0: every bank row, 1: one row, 2: sixteen rows)"
for sym in 0 1; do for part in 0 1 2; do
  run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_SYM=$sym RAISR_HIP_AC_PART=$part" --steps 4
done; done
for sym in 0 1; do for pat in 1 2; do
  run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_SYM=$sym RAISR_HIP_AC_PART=2 RAISR_HIP_AC_PATTERN=$pat" --steps 4
done; done
} 2>&1 | tee $O/ab.log
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 2 --warmup 1 --lanes 1 --frames-per-step 8"
for sym in 0 1; do
  i=0
  for grp in "TA_TA_BUSY_sum TA_BUSY_max TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    RAISR_HIP_SYM=$sym rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $R/$O/sym$sym/g$i -- $P > $R/$O/sym$sym.g$i.log 2>&1
  done
  python $R/scripts/pmc_summarize.py $R/$O/sym$sym > $R/$O/pmc_sym$sym.txt 2>&1
  echo "== counters SYM=$sym"; grep -A40 "k_hashfilter_ac" $R/$O/pmc_sym$sym.txt | head -34
done
rm -rf $R/$O/sym0 $R/$O/sym1
