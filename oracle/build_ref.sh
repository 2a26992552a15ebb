#!/usr/bin/env bash
# build_ref.sh — compile the UNMODIFIED reference aligner (src/aligner/*.cpp, where the files lie
# under /root/reference) together with our own oracle/ref_driver.cpp into oracle/_ref/.
#
# TEST INFRASTRUCTURE ONLY.  Does not run the reference's CMake build.  The only third-party
# dependency, Eigen 3.3.7, is vendored INSIDE the reference as cmake/eigen-3.3.7.zip (MD5 pinned by
# cmake/GetEigen.cmake:8); its headers are unpacked to a temporary directory under oracle/_ref/
# for the duration of the compile and removed afterwards.  No reference source is copied into
# the repository; oracle/_ref/ is git-ignored (binaries only).
#
# Outputs:
#   oracle/_ref/ref_driver       serial build  -> the parity oracle
#   oracle/_ref/ref_driver_omp   -DUSEOMP build -> timing only (racy by design)
set -euo pipefail
REF=${REFERENCE_ROOT:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src/aligner" ]; then
  echo "build_ref.sh: $REF not present; keeping prebuilt oracle/_ref (if any)" >&2
  exit 0
fi
mkdir -p "$OUT"
if [ -x "$OUT/ref_driver" ] && [ -x "$OUT/ref_driver_omp" ] && [ "$OUT/ref_driver" -nt "$HERE/ref_driver.cpp" ] \
   && [ "$OUT/ref_driver" -nt "$HERE/build_ref.sh" ]; then
  exit 0
fi
TMP="$OUT/.eigen_tmp"
rm -rf "$TMP"; mkdir -p "$TMP"
trap 'rm -rf "$TMP"' EXIT
python3 - "$REF/cmake/eigen-3.3.7.zip" "$TMP" <<'PY'
import sys, zipfile
z = zipfile.ZipFile(sys.argv[1])
for n in z.namelist():
    parts = n.split('/', 1)
    if len(parts) == 2 and parts[1].startswith('Eigen/') and not n.endswith('/'):
        z.extract(n, sys.argv[2])
PY
EIGEN="$(echo "$TMP"/eigen-eigen-*)"
SRC="$REF/src/aligner"
# the reference's own flags (CMakeLists.txt:9) minus -march=native/-flto so the binary also runs
# on the GPU box's host CPU; -mavx2 is what the _mm256_*_epi8 intrinsics need.
FLAGS="-std=c++17 -Ofast -mavx2 -mfma -ffast-math -w -I$SRC -I$EIGEN"
g++ $FLAGS "$HERE/ref_driver.cpp" "$SRC/similaritymatrix.cpp" "$SRC/smithwaterman.cpp" "$SRC/plocalaligner.cpp" \
    -o "$OUT/ref_driver"
g++ $FLAGS -DUSEOMP -fopenmp "$HERE/ref_driver.cpp" "$SRC/similaritymatrix.cpp" "$SRC/smithwaterman.cpp" \
    "$SRC/plocalaligner.cpp" -o "$OUT/ref_driver_omp"
echo "built $OUT/ref_driver $OUT/ref_driver_omp"
