#!/bin/bash
# usage (on the GPU box): scripts/pmc_probe.sh <tag> [ENV=VAL ...]   -> gpurun_out/pmc_<tag>.txt
# SQ / LDS / memory counters of the hot kernels for one configuration (separate rocprofv3 passes, --kernel-trace only).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcprobe_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
B="python $ROOT/bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 2 --warmup 1 --lanes 1 --frames-per-step 8"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/g$i" -- $B > "$OUT/g$i.log" 2>&1
done
python $ROOT/scripts/pmc_summarize.py "$OUT" > "$ROOT/gpurun_out/pmc_$TAG.txt"
cat "$ROOT/gpurun_out/pmc_$TAG.txt"
