#!/bin/bash
# GPU box: throughput of the pipeline variants with 4 and 8 frames in flight (do complementary kernels of different frames co-schedule?)
for lanes in 4 8; do
for env in "" "RAISR_HIP_SPLIT=1" "RAISR_HIP_SPLIT=1 RAISR_HIP_LDS_FILTER=0"; do
  env $env python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 --lanes $lanes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $lanes [$env]', d['value'], d['config']['fps'], d['kernels_isolated_ms'])"
done; done
