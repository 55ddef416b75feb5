#!/bin/bash
# GPU box: rocprofv3 kernel + memory-copy trace of the stream ring at a given depth -> gpurun_out/prof_stream_d<depth>
D=${1:-4}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
sed "s/for depth in .*/for depth in ($D,):/" $ROOT/scripts/stream_probe.py > /tmp/sp.py
sed -i "s#ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))#ROOT = '$ROOT'#" /tmp/sp.py
rm -rf $ROOT/gpurun_out/prof_stream_d$D
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/gpurun_out/prof_stream_d$D -- python /tmp/sp.py 300 2>&1 | grep depth
