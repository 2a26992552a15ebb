#!/usr/bin/env python3
"""Host-phase trace (MI355_SW_TRACE=1) of one warmed-up single alignment call."""
import os
import sys

os.environ["MI355_SW_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pgs = g._load_package()
ctx = pgs.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sem = int(sys.argv[2]) if len(sys.argv) > 2 else 0
refa = pgs.synth.dna(1, n)
ref = refa.tobytes()
reads = [pgs.synth.read_from_ref(refa, 2 + k, 150)[0].tobytes() for k in range(4)]
for k in range(4):
    sys.stderr.write("---- call %d\n" % k)
    sys.stderr.flush()
    ctx.align(reads[k], ref, sem)
ctx.close()
