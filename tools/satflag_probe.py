#!/usr/bin/env python3
"""Probe of the float engine's saturating sweep: flagged sub-chunk counts (MI355_SW_TRACE=1) and the kernel that ran."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pgs = g._load_package()
ctx = pgs.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 450_000
poly = b"A" * n
pa = [b"A" * 800, b"A" * 750 + b"C" + b"A" * 20, b"A" * 700]
got = ctx.align_batch(pa, poly, semantics=0)
print(ctx.last_kernel()["name"], [(r["score"], r["pos"]) for r in got])
