#!/bin/bash
# round 5, final session: the whole GPU suite + smoke on the final tree, per-configuration profiles (PMC traffic for the final kernel
# sources), default bench.py run
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_final; mkdir -p $O
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 | tee $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
for cfg in C2 C1 C3 C4 C5; do
  timeout 900 bash scripts/profile_gpu.sh r05_$cfg --config $cfg > $O/profile_$cfg.log 2>&1
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
RAISR_BENCH_DEVICES=0,0 timeout 600 python bench.py --gpus 2 --single-process --steps 2 --warmup 1 > $O/bench_single_process_2slots.json 2>> $O/bench.err; tail -c 400 $O/bench_single_process_2slots.json
timeout 600 python bench.py --gpus 1 --single-process --steps 2 --warmup 1 > $O/bench_single_process_1.json 2>> $O/bench.err; tail -c 300 $O/bench_single_process_1.json
