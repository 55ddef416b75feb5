#!/bin/bash
# (GPU box) what binds the C2 pipeline -- the power budget or the issue rate?  The four-lane device loop (frames resident in HBM) is run
#   1. as is;  2. with the shader clock capped (rocm-smi --setperfdeterminism <MHz>: 2100, 1900, 1700);  3. with the socket power capped
#   (rocm-smi --setpoweroverdrive <W>: -10 %, -20 % of the board limit), each for ~20 s, sampling clock and power under load.
# Reading: fps falling 1:1 with the clock cap and NOT (or less than proportionally) with the power cap = issue-bound, the power lead is
# closed; fps falling with the power cap while the clock cap at the same resulting clock costs the same = power-bound.
# Every setting is reset at the end (and on any exit).   usage: scripts/power_cap_probe.sh   -> gpurun_out/power_cap/log.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/power_cap; mkdir -p $O; : > $O/log.txt
reset_all() { rocm-smi --resetperfdeterminism >/dev/null 2>&1; rocm-smi --resetpoweroverdrive >/dev/null 2>&1; rocm-smi --resetclocks >/dev/null 2>&1; }
trap reset_all EXIT
sample() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics" | sed -e 's/^GPU\[0\]\s*: //' | tr '\n' ';'; }
run() {   # label
  python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 50 --warmup 3 > $O/bench_$1.json 2>/dev/null &
  B=$!
  sleep 5
  for i in 1 2 3 4; do echo "  [$1] load: $(sample)" >> $O/log.txt; sleep 1.0; done
  wait $B
  python - "$1" "$O/bench_$1.json" <<'P' | tee -a $O/log.txt
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print(f"{sys.argv[1]}: fps {d['config']['fps']:.0f}")
except Exception as e:
    print(f"{sys.argv[1]}: no result ({e})")
P
}
echo "limits: $(rocm-smi --showmaxpower 2>/dev/null | grep -i 'max' | tr '\n' ';') $(rocm-smi --showpowerprofile 2>/dev/null | head -0)" | tee -a $O/log.txt
echo "idle: $(sample)" | tee -a $O/log.txt
run baseline
for mhz in 2100 1900 1700; do
  out=$(rocm-smi --setperfdeterminism $mhz 2>&1 | grep -iE "success|fail|error|not|set" | head -2 | tr '\n' ';')
  echo "setperfdeterminism $mhz: $out" | tee -a $O/log.txt
  run clk$mhz
  rocm-smi --resetperfdeterminism >/dev/null 2>&1
done
MAXW=$(rocm-smi --showmaxpower 2>/dev/null | grep -oE "[0-9]+(\.[0-9]+)? *W?$" | grep -oE "^[0-9]+" | head -1)
[ -z "$MAXW" ] && MAXW=1400
for pct in 90 80 70; do
  w=$(( MAXW * pct / 100 ))
  out=$(rocm-smi --setpoweroverdrive $w --autorespond y 2>&1 | grep -iE "success|fail|error|not|set|invalid" | head -2 | tr '\n' ';')
  echo "setpoweroverdrive ${w} W (${pct} % of $MAXW): $out" | tee -a $O/log.txt
  run pw$pct
  rocm-smi --resetpoweroverdrive >/dev/null 2>&1
done
reset_all
run baseline_again
cat $O/log.txt | grep -v "^  \[" 
