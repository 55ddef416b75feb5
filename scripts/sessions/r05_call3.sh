#!/bin/bash
# round 5, call 3: binary16 pair windows with array B padded (LDS bank conflicts): parity, A/B against the round-4 library, LDS counters;
# then the ceiling of a symmetric filter stage on the banks that are not symmetric (timing probe of a development build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_call3; mkdir -p $O
D=video-super-resolution-library_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_batch.py -q -x -m gpu -k "fp16 or C4 or 16" 2>&1 | tail -5 | tee $O/parity.log
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['fps'], d['kernels_isolated_ms'])"; }
{
echo "== C4"
for rep in 1 2 3; do
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_r04.so --config C4
run X=pad --config C4
done
for cfg in C5 C3 C1; do
echo "== $cfg (development build)"
for rep in 1 2; do
run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so" --config $cfg
run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_SYM_IGNORE_ASYM=1" --config $cfg
done; done
run "RAISR_HIP_LIB=$R/$D/_exp/libraisr_dev.so RAISR_HIP_SYM_MAX_ROWS=64" --config C5
echo "== C2 sanity (round-4 library vs tree)"
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_r04.so --config C2
run X=tree --config C2
run RAISR_HIP_LIB=$R/$D/_exp/libraisr_r04.so --config C2
run X=tree --config C2
} 2>&1 | tee $O/ab.log
# LDS counters of k_hashfilter16, round-4 library and tree
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 2 --warmup 1 --lanes 1 --frames-per-step 8 --config C4"
for v in r04 tree; do
  [ $v = r04 ] && export RAISR_HIP_LIB=$R/$D/_exp/libraisr_r04.so || unset RAISR_HIP_LIB
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $R/$O/pmc_$v -- $B > $R/$O/pmc_$v.log 2>&1
  python - <<PY | tee $R/$O/pmc_$v.txt
import csv,glob,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("$R/$O/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "hashfilter16" not in k: continue
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,d in tot.items():
    n=cnt[(k,"SQ_LDS_IDX_ACTIVE")]
    print("$v", k[:60], "launches", n, {c: round(v/n) for c,v in d.items()})
PY
done
