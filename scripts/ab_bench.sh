#!/bin/bash
# A/B on ONE GPU box: bench.py with the in-tree library and with every candidate library under
# video-super-resolution-library_amd/_exp/libraisr_<name>.so (built on the CPU side), base first and last.
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
cp $D/libraisr_hip.so /tmp/base.so
run() { echo -n "$1: "; python bench.py --no-cpu-baseline --steps 20 --warmup 3 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
run base "$@"
for f in $D/_exp/libraisr_*.so; do
  [ -e "$f" ] || continue
  cp "$f" $D/libraisr_hip.so; n=$(basename "$f" .so); run "${n#libraisr_}" "$@"
done
cp /tmp/base.so $D/libraisr_hip.so
run base "$@"
