#!/usr/bin/env python3
"""One rank's LPT share of config 4 at a given world size on one GPU, with the host phases traced (option trace -> stderr):
usage: c4_share_trace.py [world] [score_only]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pgs = bench.load_package()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
score_only = len(sys.argv) > 2 and sys.argv[2] == "1"
dist = importlib.import_module("parallel_genomeseq_amd.dist")
nseq = 561_356
lens = pgs.synth.lognormal_lengths(5, nseq)
allres = pgs.synth.protein(5, int(lens.sum()))
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
w = lens.astype(np.float64) * len(pgs.synth.P02232)
parts = dist.shard_lpt(w, world) if world > 1 else [np.arange(nseq, dtype=np.int64)]
mine = parts[int(np.argmax([float(w[p].sum()) for p in parts]))]
sl = lens[mine]
so = np.concatenate([[0], np.cumsum(sl)]).astype(np.int64)
buf = np.empty(int(sl.sum()), dtype=np.uint8)
for k, i in enumerate(mine):
    buf[so[k]:so[k + 1]] = allres[offs[i]:offs[i + 1]]
ctx = pgs.Context(0)
ctx.set_reference(pgs.synth.P02232)
ctx.batch_upload_packed(buf, so)
flags = pgs.capi.SCORE_ONLY if score_only else 0
for _ in range(3):
    ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
ctx.set_option("trace")
t0 = time.perf_counter()
ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
dt = time.perf_counter() - t0
ctx.set_option("trace", None)
print("world %d: %d sequences, longest %d, call %.3f ms, device %.3f ms, path %s" % (world, len(mine), int(sl.max()), dt * 1e3,
                                                                                  ctx.last_timings()["total_us"] * 1e-3, " ".join(ctx.last_path())))
ctx.close()
