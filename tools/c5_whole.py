#!/usr/bin/env python3
"""Config 5 end to end on one GPU (10 kbp x 250 Mbp, float engine): score pass, locate, traceback — twice.  For profiling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pgs = bench.load_package()
n, m = 250_000_000, 10_000
q, off = bench.config5_inputs(pgs, n, m)
ref = pgs.synth.dna(6, n)
ctx = pgs.Context(0)
ctx.set_reference(ref); ctx.batch_upload([q])
for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    ctx.set_option(k, v or True)
for _ in range(2):
    t0 = time.perf_counter(); r = ctx.batch_run(semantics=0)[0]; dt = time.perf_counter() - t0
    tm = ctx.last_timings()
    print("%.1f ms call: score %.1f locate %.1f traceback %.1f ms; %s; score %g pos %d (planted %d)" %
          (dt * 1e3, tm["score_us"] / 1e3, tm["locate_us"] / 1e3, tm["trace_us"] / 1e3, ctx.last_kernel()["name"], r["score"], r["pos"], off + 1), file=sys.stderr)
