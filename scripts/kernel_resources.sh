#!/bin/bash
# (build container) registers / scratch / LDS of the kernels, from a device-only compile of csrc/device_abi.hip:
#   scripts/kernel_resources.sh [name-filter] [-DMACRO ...]      (keeps the assembly in /tmp/raisr_dev.s)
cd "$(dirname "$0")/../video-super-resolution-library_amd"
flt=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable -Wno-pass-failed "$@" \
    --cuda-device-only -S -o /tmp/raisr_dev.s csrc/device_abi.hip || exit 1
python3 - "$flt" <<'PY'
import re,sys
txt=open('/tmp/raisr_dev.s').read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
meta=txt[txt.index('amdhsa.kernels:'):]
for blk in meta.split('  - .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    n=g('name')
    if flt in n:
        print(f"{n[:100]:100s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6} vspill {g('vgpr_spill_count')}")
PY
