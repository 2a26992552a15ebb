#!/usr/bin/env python3
"""Config 4 on one GPU (561 356 UniProt-shaped sequences vs P02232) through the packed upload: score + argmax, and with
traceback, three times each.  For profiling."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pgs = bench.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 561_356
lens = pgs.synth.lognormal_lengths(5, n)
allres = pgs.synth.protein(5, int(lens.sum()))
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
ctx = pgs.Context(0)
if os.environ.get("C4_TRACE"): ctx.set_option("trace", 1)
ctx.set_reference(pgs.synth.P02232)
t0 = time.perf_counter(); ctx.batch_upload_packed(allres, offs); print("upload %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
for flags in (pgs.capi.SCORE_ONLY, 0):
    for _ in range(3):
        t0 = time.perf_counter(); ctx.batch_run(semantics=0, flags=flags, raw=True); dt = time.perf_counter() - t0
    print("flags %d: %.1f ms wall, %.1f ms device" % (flags, dt * 1e3, ctx.last_timings()["total_us"] / 1e3), file=sys.stderr)
