#!/bin/bash
# throughput per frame kind (content dependence of the certified hash stage) and per BASELINE config
B="python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 4 --warmup 1 --frames-per-step 256"
for k in natural random constant checker; do echo "C2 $k: $($B --frame-kind $k 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["config"]["fps"], "fps", j["value"], "MP/s")')"; done
for c in C1 C3 C4 C5; do echo "$c natural: $($B --config $c 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.readline()); print(j["config"]["fps"], "fps", j["value"], "MP/s")')"; done
