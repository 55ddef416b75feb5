#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# usage: scripts/profile_gpu.sh <tag> [--config Cn] [ENV=VAL ...]     e.g.  scripts/profile_gpu.sh r03_C3 --config C3   |   scripts/profile_gpu.sh r03_split RAISR_HIP_SPLIT=1
# Output goes to gpurun_out/prof_<tag>/ ; summarise afterwards (in the build container, same sources) with scripts/summarize_profiles.py <tag>.
# PMC passes are separate runs with --kernel-trace only (gpurun refuses pmc + sys-trace combos).
set -u
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
echo "$*" > "$OUT/variant.txt"
BARGS=""
for kv in "$@"; do case "$kv" in --*|C[1-5]|C[1-5][a-z]|[0-9]*) BARGS="$BARGS $kv";; *) export "$kv";; esac; done      # "--config C3" goes to bench.py, KEY=VAL to the environment
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-extras$BARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B --steps 3 --warmup 1 > "$OUT/stats.log" 2>&1
$B --steps 3 --warmup 1 > "$OUT/bench_events.json" 2> "$OUT/bench_events.err"
PMC="$B --steps 2 --warmup 1 --lanes 1 --no-kernel-timing --frames-per-step 8"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS -d "$OUT/pmc_sq" -- $PMC > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d "$OUT/pmc_lds" -- $PMC > "$OUT/pmc_lds.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- $PMC > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -- $PMC > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -- $PMC > "$OUT/pmc_l2.log" 2>&1
# matrix-core counters (only the opt-in fast mode issues MFMAs)
[ -n "${RAISR_HIP_FAST:-}" ] && rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES -d "$OUT/pmc_mfma" -- $PMC > "$OUT/pmc_mfma.log" 2>&1
echo done
