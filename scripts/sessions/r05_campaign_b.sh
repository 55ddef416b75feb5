#!/bin/bash
# round 5, campaign rerun after the coherence bound changed (R5.7): both frame-family sets at full size
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_campaign_b; mkdir -p $O
KINDS=r05 timeout 1500 python scripts/certify_campaign.py 30 > $O/campaign_r05.log 2>&1; tail -3 $O/campaign_r05.log
cp gpurun_out/certify_campaign.json $O/certify_campaign_r05b_new_families.json 2>/dev/null
timeout 1500 python scripts/certify_campaign.py 30 > $O/campaign_base.log 2>&1; tail -3 $O/campaign_base.log
cp gpurun_out/certify_campaign.json $O/certify_campaign_r05b_base_families.json 2>/dev/null
