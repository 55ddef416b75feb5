#!/bin/bash
# GPU session 8 of round 3: C4 after the group-wise fold, texture-path busy counters of the fused kernel, e2e trace of the chunked plugin path
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_call8; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "fp16 or C4 or c4" > $O/fp16_tests.log 2>&1; tail -3 $O/fp16_tests.log
B="python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['fps'], d['kernels_isolated_ms'], d['kernels_avg_ms'])"; }
{ echo "== C4"; $B --config C4 2>/dev/null | show; echo "== C4 lanes 8"; $B --config C4 --lanes 8 2>/dev/null | show; } > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 2 --warmup 1 --lanes 1 --frames-per-step 8"
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_max TA_BUSY_min GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $R/$O/ta$i -- $P > $R/$O/ta$i.log 2>&1
done
python $R/scripts/pmc_summarize.py $R/$O/ta1 $R/$O/ta2 $R/$O/ta3 $R/$O/ta4 $R/$O/ta5 > $R/$O/ta.txt 2>&1; cat $R/$O/ta.txt | head -60
RAISR_HIP_CHUNKS=3 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_chunks3 -- python $R/scripts/e2e_probe.py > $R/$O/trace3.log 2>&1
RAISR_HIP_CHUNKS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_chunks1 -- python $R/scripts/e2e_probe.py > $R/$O/trace1.log 2>&1
ls $R/$O/trace_chunks3/*/ | head
echo done
