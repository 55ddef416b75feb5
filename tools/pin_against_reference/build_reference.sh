#!/bin/bash
# Builds the UNMODIFIED reference library (OpenVisualCloud/Video-Super-Resolution-Library) twice, as shared objects that
# run_reference.py loads:
#   libraisr_ref_strict.so    -O3 -march=native -fno-fast-math -ffp-contract=off  (the normative semantics: every intrinsic one IEEE op)
#   libraisr_ref_shipped.so   the flags of the reference's CMakeLists.txt:23-41 (-ffast-math), i.e. what FFmpeg links
# both with -DUSE_ATAN2_APPROX (what CMake selects for every compiler but icpx).
#
# Needs what this repository's build container lacks and may not fake: Intel IPP (oneAPI 2021.12 / 2022.0, the versions the
# reference's scripts and CI install) and, for the bits to be the reference's, an Intel CPU with AVX-512
# (VRCP14PS / VRSQRT14PS differ between vendors).  Nothing of the reference is copied: it is compiled where it lies.
#
# usage: build_reference.sh <reference checkout> <IPPROOT> <output dir> [CXX]
set -euo pipefail
REF=${1:?reference checkout}; IPP=${2:?IPPROOT (contains include/ipp.h)}; OUT=${3:?output dir}; CXX=${4:-g++}
[ -f "$REF/Library/Raisr.cpp" ] || { echo "no Library/Raisr.cpp under $REF" >&2; exit 1; }
[ -f "$IPP/include/ipp.h" ] || [ -f "$IPP/include/ipp/ipp.h" ] || { echo "no ipp.h under $IPP/include -- this kit does not ship a stand-in" >&2; exit 1; }
INC="-I$IPP/include"; [ -f "$IPP/include/ipp/ipp.h" ] && INC="$INC -I$IPP/include/ipp"
LIBDIR=$IPP/lib; [ -d "$IPP/lib/intel64" ] && LIBDIR=$IPP/lib/intel64
mkdir -p "$OUT"
COMMON="-std=c++17 -O3 -march=native -DNDEBUG -DUSE_ATAN2_APPROX -Wno-narrowing -fPIC $INC -I$REF/Library"
# Raisr.cpp textually includes Raisr_AVX256.cpp / Raisr_AVX512.cpp / Raisr_AVX512FP16.cpp (Raisr.cpp:11,27,31): two translation units
build() {   # name, extra flags
  for tu in Raisr RaisrHandler; do
    $CXX $COMMON $2 -c "$REF/Library/$tu.cpp" -o "$OUT/$1_$tu.o"
  done
  # the driver is Python (ctypes): no executable is linked with -ffast-math, so crtfastmath.o never sets FTZ/DAZ
  $CXX -shared -o "$OUT/libraisr_ref_$1.so" "$OUT/$1_Raisr.o" "$OUT/$1_RaisrHandler.o" \
       -L"$LIBDIR" -Wl,-rpath,"$LIBDIR" -lippi -lipps -lippvm -lippcore -lpthread
  rm -f "$OUT/$1_Raisr.o" "$OUT/$1_RaisrHandler.o"
}
build strict  "-fno-fast-math -ffp-contract=off"
build shipped "-ffast-math"
{
  echo "compiler: $($CXX --version | head -1)"
  echo "reference_commit: $(git -C "$REF" rev-parse HEAD 2>/dev/null || echo unknown)"
  echo "ipp_root: $IPP"
  grep -m1 "model name" /proc/cpuinfo | sed 's/.*: /cpu: /'
  grep -m1 -o "avx512_fp16" /proc/cpuinfo | sed 's/^/cpu_has: /' || true
} > "$OUT/build_info.txt"
cat "$OUT/build_info.txt"
echo "built $OUT/libraisr_ref_strict.so and $OUT/libraisr_ref_shipped.so"
