#!/usr/bin/env python3
"""Config 5 (10 kbp x 250 Mbp, float engine) under a few option sets: score kernel / locate / traceback ms, counters.
usage: c5_ab.py [ref_len] [query_len] [opt=value,opt=value;...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pgs = bench.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
sets = sys.argv[3].split(";") if len(sys.argv) > 3 else ["", "no_long_save=1"]
q, off = bench.config5_inputs(pgs, n, m)
ref = pgs.synth.dna(6, n)
ctx = pgs.Context(0)
ctx.set_reference(ref)
ctx.batch_upload([q])
for s in sets:
    opts = dict(kv.split("=") for kv in s.split(",") if kv)
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.batch_run(semantics=pgs.F32)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        r = ctx.batch_run(semantics=pgs.F32)[0]
        dt = time.perf_counter() - t0
        tm = ctx.last_timings()
        if best is None or dt < best[0]:
            best = (dt, tm)
    print(json.dumps({"options": opts, "wall_ms": best[0] * 1e3, "score_kernel_ms": best[1]["score_us"] * 1e-3, "locate_ms": best[1]["locate_us"] * 1e-3,
                      "traceback_ms": best[1]["trace_us"] * 1e-3, "score": r["score"], "pos": r["pos"], "counters": ctx.last_counters(),
                      "kernel": ctx.last_kernel()["name"][:100]}), flush=True)
    for k in opts:
        ctx.set_option(k, None)
ctx.close()
