#!/bin/bash
# A/B on ONE GPU box through RAISR_HIP_LIB (no file is swapped): the in-tree library against candidate / previous builds under
# video-super-resolution-library_amd/_exp/libraisr_<name>.so, interleaved `reps` times so that clock drift hits both sides alike.
#   usage: scripts/ab_lib_bench.sh "<names...>" <reps> [bench.py args]        e.g.  ab_lib_bench.sh "prev" 3 --config C2
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=$PWD/video-super-resolution-library_amd
names=$1; reps=$2; shift 2
run() {  # label, lib ("" = in-tree)
  echo -n "$1: "
  if [ -n "$2" ]; then export RAISR_HIP_LIB=$2; else unset RAISR_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 "${@:3}" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], {k: round(v,4) for k,v in d['kernels_isolated_ms'].items()})"
  unset RAISR_HIP_LIB
}
for r in $(seq $reps); do
  run "new " "" "$@"
  for n in $names; do run "$n" "$D/_exp/libraisr_$n.so" "$@"; done
done
