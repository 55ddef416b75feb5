#!/bin/bash
# round 5, evidence session: per-configuration rocprofv3 profiles (kernel trace + PMC passes), counters of the deferred comparison
# pipeline against the in-tile worklist, the certification campaign on three new frame families, and a default bench.py run
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r05_evidence; mkdir -p $O
D=video-super-resolution-library_amd
for cfg in C2 C1 C3 C4 C5; do
  timeout 900 bash scripts/profile_gpu.sh r05_$cfg --config $cfg > $O/profile_$cfg.log 2>&1
done
timeout 600 bash scripts/pmc_probe.sh r05_intile > $O/pmc_intile.log 2>&1
timeout 600 bash scripts/pmc_probe.sh r05_defer RAISR_HIP_LIB=$R/$D/libraisr_hip_testhooks.so RAISR_HIP_DEFER=1 > $O/pmc_defer.log 2>&1
KINDS=r05 timeout 1500 python scripts/certify_campaign.py 30 > $O/campaign.log 2>&1; tail -3 $O/campaign.log
cp gpurun_out/certify_campaign.json $O/certify_campaign_r05.json 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
