#!/usr/bin/env bash
# build_dropin.sh — compile tests/cpp/test_multi.cpp (C-ABI, mi355_sw_multi_*) and tests/cpp/test_dropin.cpp against
# include/parseq/*.h, the latter twice:
#   test_dropin.bin        without Eigen on the include path (parseq::Timings / parseq::DenseMatrix signatures)
#   test_dropin_eigen.bin  with Eigen 3.3.7 (the reference's own vendored zip, cmake/eigen-3.3.7.zip, unpacked to a
#                          temporary directory for the duration of the compile and removed afterwards): the reference's
#                          signatures — Eigen::VectorXf getTimings(), const Eigen::MatrixXf &get_matrix(), MatrixX8u.
# The second binary is only built where /root/reference exists; both are git-ignored and travel to the GPU box.
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
fi
