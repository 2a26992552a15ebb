#!/usr/bin/env python3
"""Latency of OMPParallelLocalAligner-style calls (mi355_sw_align_split), the loop of src/sw_solve_big.cpp:78-92:
one 150 bp read, npiece pieces, overlap 2.0, same reference every call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pgs = g._load_package()
ctx = pgs.Context(0)
n = int(sys.argv[1]); sem = int(sys.argv[2]); npiece = int(sys.argv[3]) if len(sys.argv) > 3 else 16
refa = pgs.synth.dna(1, n); ref = refa.tobytes()
reads = [pgs.synth.read_from_ref(refa, 2 + k, 150)[0].tobytes() for k in range(8)]
for k in range(4): ctx.align_split(reads[k], ref, npiece, 2.0, sem, sem)
t0 = time.perf_counter()
for k in range(64): ctx.align_split(reads[k % 8], ref, npiece, 2.0, sem, sem)
print("split x%d: ms per call %.3f" % (npiece, (time.perf_counter() - t0) / 64 * 1e3), file=sys.stderr)
