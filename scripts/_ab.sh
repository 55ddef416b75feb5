cd ${GRAFT_REPO_ROOT:-/root/repo}
D=video-super-resolution-library_amd
cp $D/libraisr_hip.so /tmp/base.so
for v in base u8 u16 base; do
  if [ $v = base ]; then cp /tmp/base.so $D/libraisr_hip.so; else cp $D/_exp/libraisr_$v.so $D/libraisr_hip.so; fi
  touch $D/libraisr_hip.so
  echo -n "$v: "; python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"
done
cp /tmp/base.so $D/libraisr_hip.so
