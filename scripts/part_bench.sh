#!/bin/bash
# isolated kernel times of the certified fused kernel and its parts (profiling aid)
B="python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --frames-per-step 96"
for part in 0 1 2; do
  echo "AC_PART=$part: $(RAISR_HIP_AC_PART=$part $B 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["kernels_isolated_ms"], j["config"]["fps"])')"
done
echo "exact: $(RAISR_HIP_CERTIFY=0 $B 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["kernels_isolated_ms"], j["config"]["fps"])')"
echo "exact unfused: $(RAISR_HIP_CERTIFY=0 RAISR_HIP_FUSED=0 $B 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["kernels_isolated_ms"], j["config"]["fps"])')"
echo "split_ac: $(RAISR_HIP_SPLIT=1 $B 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["kernels_isolated_ms"], j["config"]["fps"])')"
