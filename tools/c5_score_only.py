#!/usr/bin/env python3
"""Config 5's score pass alone (10 kbp x 250 Mbp, float engine, whole reference), twice per option set: for rocprofv3 / PMC
passes and A/B runs.  usage: c5_score_only.py [opt=val,opt=val ...]   (each argument = one set of context options)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pgs = bench.load_package()
n, m = 250_000_000, 10_000
q, off = bench.config5_inputs(pgs, n, m)
ref = pgs.synth.dna(6, n)
ctx = pgs.Context(0)
ctx.set_reference(ref); ctx.batch_upload([q])
for cfg in (sys.argv[1:] or [""]):
    opts = [(kv.split("=") + ["1"])[:2] for kv in cfg.split(",") if kv]
    for k, v in opts:
        ctx.set_option(k, v)
    for sem in (0, 1):
        for _ in range(2):
            t0 = time.perf_counter(); r = ctx.batch_run(semantics=sem, flags=pgs.capi.SCORE_ONLY)[0]; dt = time.perf_counter() - t0
            tm = ctx.last_timings(); ki = ctx.last_kernel()
        print("[%s] sem %d: %.1f ms score kernel, %.1f ms call, %s chunk %d sub %d warm %d score %g end_y %d" % (cfg, sem, tm["score_us"] / 1e3, dt * 1e3, ki["name"], ki["chunk_len"], ki["sub_len"], ki["warm"], r["score"], r["end_y"]), file=sys.stderr, flush=True)
        if os.environ.get("C5_F32_ONLY"):
            break
    for k, v in opts:
        ctx.set_option(k, None)
