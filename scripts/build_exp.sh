#!/bin/bash
# (build container) candidate library for scripts/ab_bench.sh:  scripts/build_exp.sh <name> [-DFLAG ...]
# -> video-super-resolution-library_amd/_exp/libraisr_<name>.so, same flags as the product Makefile plus the given ones.
set -e
cd "$(dirname "$0")/../video-super-resolution-library_amd"
name=$1; shift
mkdir -p _exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function "$@" \
    -shared -o _exp/libraisr_$name.so csrc/device_abi.hip csrc/raisr_api.cpp csrc/raisr_stream.cpp -ldl
echo "built _exp/libraisr_$name.so ($*)"
