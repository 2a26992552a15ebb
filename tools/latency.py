#!/usr/bin/env python3
"""Latency of single-alignment calls (the one-by-one loop of the reference's drivers): mi355_sw_align of one
150 bp read against 1 Mbp and 50 Mbp references, same reference every call (kept resident by content hash)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pgs = g._load_package()
ctx = pgs.Context(0)
for n in (1_000_000, 50_000_000):
    ref = pgs.synth.dna(1, n).tobytes()
    reads = [pgs.synth.read_from_ref(pgs.synth.dna(1, n), 2 + k, 150)[0].tobytes() for k in range(8)]
    for sem, name in ((0, "f32"), (1, "u8")):
        ctx.align(reads[0], ref, sem)
        t0 = time.perf_counter()
        for k in range(16):
            ctx.align(reads[k % 8], ref, sem)
        dt = (time.perf_counter() - t0) / 16
        tm = ctx.last_timings()
        print("%s ref %d: %.2f ms per align (device total %.2f ms, score kernel %.2f ms) = %.1f GCUPS" %
              (name, n, dt * 1e3, tm["total_us"] / 1e3, tm["score_us"] / 1e3, 150 * n / dt * 1e-9), flush=True)
ctx.close()
