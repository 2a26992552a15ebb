#!/usr/bin/env python3
"""One-by-one calls, short reads: ms per call by read length, reference length and engine (A/B: MI355_SW_NO_COMB=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pgs = g._load_package()
ctx = pgs.Context(0)
for n in ([int(a) for a in sys.argv[1:]] or [1_000_000, 50_000_000]):
    refa = pgs.synth.dna(1, n); ref = refa.tobytes()
    for m in (100, 150, 250, 400):
        reads = [pgs.synth.read_from_ref(refa, 2 + k, m)[0].tobytes() for k in range(8)]
        for sem in (0, 1):
            for k in range(3): ctx.align(reads[k], ref, sem)
            t0 = time.perf_counter()
            reps = 32
            for k in range(reps): ctx.align(reads[k % 8], ref, sem)
            dt = (time.perf_counter() - t0) / reps
            t = ctx.last_timings()
            print("n=%d len=%d sem=%d: %.3f ms per call (score %.3f) %.2f TCUPS | %s" % (n, m, sem, dt * 1e3, t["score_us"] * 1e-3, m * n / dt * 1e-12, ctx.last_kernel()["name"][:70]), flush=True)
