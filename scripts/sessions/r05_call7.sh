#!/bin/bash
# round 5, call 7: the whole GPU suite on the restructured libraries (product / test-hooks / development flavours, E_ay with both roundings)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_call7; mkdir -p $O
timeout 3300 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 | tee $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
