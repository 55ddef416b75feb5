#!/bin/bash
# per-phase shader-clock cycles of the certified hash stage (wave 0, summed over tiles), one lane, 16 frames
B="python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 1 --warmup 0 --frames-per-step 16 --lanes 1"
for part in 0 1; do echo "PART=$part"; RAISR_HIP_PHASES=1 RAISR_HIP_AC_PART=$part $B 2>&1 | grep phases; done
echo "SPLIT"; RAISR_HIP_PHASES=1 RAISR_HIP_SPLIT=1 $B 2>&1 | grep phases
