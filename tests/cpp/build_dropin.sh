#!/usr/bin/env bash
# build_dropin.sh — compile tests/cpp/test_multi.cpp (C-ABI, mi355_sw_multi_*) and tests/cpp/test_dropin.cpp against
# include/parseq/*.h, the latter twice:
#   test_dropin.bin        without Eigen on the include path (parseq::Timings / parseq::DenseMatrix signatures)
#   test_dropin_eigen.bin  with Eigen 3.3.7 (the reference's own vendored zip, cmake/eigen-3.3.7.zip, unpacked to a
#                          temporary directory for the duration of the compile and removed afterwards): the reference's
#                          signatures — Eigen::VectorXf getTimings(), const Eigen::MatrixXf &get_matrix(), MatrixX8u.
#   test_reference_gtests.bin  the reference's own test/*.cpp, unchanged and in place, against the mirror (+ gtest_shim)
# The last two are only built where /root/reference exists; all are git-ignored and travel to the GPU box.
set -euo pipefail
REF=${REFERENCE_ROOT:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
FLAGS=(-std=c++17 -O1 -Wall -Werror -pthread "-I$ROOT/include")
LINK=("-L$ROOT/parallel-genomeseq_amd" -lmi355_sw "-Wl,-rpath,$ROOT/parallel-genomeseq_amd")
g++ "${FLAGS[@]}" -DPARSEQ_NO_EIGEN "$HERE/test_dropin.cpp" "${LINK[@]}" -o "$HERE/test_dropin.bin"
g++ "${FLAGS[@]}" "$HERE/test_multi.cpp" "${LINK[@]}" -o "$HERE/test_multi.bin"
ZIP="$REF/cmake/eigen-3.3.7.zip"
if [ -f "$ZIP" ]; then
  TMP="$(mktemp -d "${TMPDIR:-/tmp}/parseq_eigen.XXXXXX")"
  trap 'rm -rf "$TMP"' EXIT
  python3 - "$ZIP" "$TMP" <<'PY'
import sys, zipfile
z = zipfile.ZipFile(sys.argv[1])
names = [n for n in z.namelist() if "/Eigen/" in n and "/unsupported/" not in n]
z.extractall(sys.argv[2], names)
PY
  INC="$(dirname "$(find "$TMP" -type d -name Eigen -not -path '*/unsupported/*' | head -1)")"
  g++ "${FLAGS[@]}" -Wno-error -w "-isystem$INC" "$HERE/test_dropin.cpp" "${LINK[@]}" -o "$HERE/test_dropin_eigen.bin"
  # the reference's OWN test files, compiled where they lie (test/main.cpp, test_localaligner.cpp, test_skewedmatrix.cpp,
  # test_OpenMP.cpp, test_MPI.cpp — the sources of its `tests` target, CMakeLists.txt:102-106) against include/parseq/
  # (their "similaritymatrix.h" / "localaligner.h" / "smithwaterman.h" resolve to the mirror) and a build-owned minimal
  # <gtest/gtest.h> (tests/cpp/gtest_shim): nothing of the reference is copied, the binary travels like oracle/_ref
  if [ -d "$REF/test" ]; then
    g++ -std=c++17 -O1 -w -pthread "-I$HERE/gtest_shim" "-I$ROOT/include/parseq" "-I$ROOT/include" "-isystem$INC" \
        "$REF/test/main.cpp" "$REF/test/test_localaligner.cpp" "$REF/test/test_skewedmatrix.cpp" "$REF/test/test_OpenMP.cpp" \
        "$REF/test/test_MPI.cpp" "${LINK[@]}" -o "$HERE/test_reference_gtests.bin"
  fi
fi
