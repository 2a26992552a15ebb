#!/usr/bin/env python3
"""config 5's score pass alone (10 kbp x 250 Mbp, float engine): ms of the score kernel under the current environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pgs = g._load_package()
n, m = 250_000_000, 10_000
ref = pgs.synth.dna(6, n)
q, off = pgs.synth.read_from_ref(ref, 7, m, sub_rate=0.01, indel_rate=0.001)
ctx = pgs.Context(0)
ctx.set_reference(ref); ctx.batch_upload([q.tobytes()])
for sem in (0, 1):
    ctx.batch_run(semantics=sem, flags=pgs.capi.SCORE_ONLY)
    t0 = time.perf_counter(); r = ctx.batch_run(semantics=sem, flags=pgs.capi.SCORE_ONLY)[0]; dt = time.perf_counter() - t0
    tm = ctx.last_timings(); ki = ctx.last_kernel()
    print("sem %d: %.1f ms score kernel, %.1f ms call, %s chunk %d warm %d score %g end_y %d" % (sem, tm["score_us"] / 1e3, dt * 1e3, ki["name"], ki["chunk_len"], ki["warm"], r["score"], r["end_y"]), file=sys.stderr)
