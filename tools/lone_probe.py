#!/usr/bin/env python3
"""One-by-one calls with longer reads: ms per call and TCUPS by read length and engine.
Usage: python tools/lone_probe.py [ref_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pgs = g._load_package()
ctx = pgs.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
refa = pgs.synth.dna(1, n); ref = refa.tobytes()
for m in (150, 400, 700, 1000, 2048, 4096):
    reads = [pgs.synth.read_from_ref(refa, 2 + k, m)[0].tobytes() for k in range(4)]
    for sem in (0, 1):
        for k in range(2): ctx.align(reads[k], ref, sem)
        t0 = time.perf_counter()
        reps = 8
        for k in range(reps): ctx.align(reads[k % 4], ref, sem)
        dt = (time.perf_counter() - t0) / reps
        t = ctx.last_timings()
        lk = ctx.last_kernel()
        print("len=%d sem=%d: %.3f ms per call, %.2f TCUPS | score %.3f locate %.3f trace %.3f ms | chunk %d sub %d warm %d | %s"
              % (m, sem, dt * 1e3, m * n / dt * 1e-12, t["score_us"] * 1e-3, t["locate_us"] * 1e-3, t["trace_us"] * 1e-3, lk["chunk_len"], lk["sub_len"], lk["warm"], lk["name"]), flush=True)
