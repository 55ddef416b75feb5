#!/bin/bash
# GPU box: isolated time of the fast mode's kernels with parts of k_filter_mfma disabled (RAISR_HIP_MF_PART: 1 = staging + sort only, 2 = conflict-free A reads)
for part in 0 1 2; do
  RAISR_HIP_FAST=1 RAISR_HIP_MF_PART=$part python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('part $part', d['value'], d['kernels_isolated_ms'])"
done
