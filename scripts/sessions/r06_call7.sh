#!/bin/bash
# round 6, session 7: (a) ceiling of a second certification level: timing probe build _exp/libraisr_l2probe.so
# (-DRAISR_PROBE_L2CERT: no 16-lane exact tensors, no barrier behind them; output wrong for ~0.02 % of the pixels) against the in-tree
# library; (b) where a wave's life goes in k_hashfilter16 (development build, phase marks added this round).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call7; mkdir -p $O
for args in "--config C2" "--config C2 --frame-kind photo" "--config C1" "--config C5 --frames-per-step 96"; do
  echo "== $args" | tee -a $O/l2probe_ab.log
  bash scripts/ab_lib_bench.sh "l2probe" 3 $args 2>&1 | tee -a $O/l2probe_ab.log
done
export RAISR_HIP_LIB=$PWD/video-super-resolution-library_amd/_exp/libraisr_dev.so
python scripts/phase_cycles.py C4 4 2>/dev/null | tee $O/phase_cycles_C4.txt
python scripts/phase_cycles.py C2 4 2>/dev/null | tee $O/phase_cycles_C2.txt
