#!/bin/bash
# round 6, session 1: real picture content through the path -- photo parity tests (C1..C5 + C2b full size, the 14 CASES at 416x240), the
# extended full-size matrix (C2b), the photo certification campaign at full size, and the default bench.py (new legs: configs.C2b,
# frame_kinds.photo).  No kernel change since round 5: this is the baseline of the round.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_call1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_photos.py tests/test_gpu_baseline_configs.py -q -x -m gpu 2>&1 | tail -6 | tee $O/photo_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
KINDS=photo timeout 1500 python scripts/certify_campaign.py 35 > $O/campaign_photo.log 2>&1; tail -3 $O/campaign_photo.log
cp gpurun_out/certify_campaign.json $O/certify_campaign_r06_photo.json 2>/dev/null
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
