#!/usr/bin/env python3
"""One rank's LPT share of config 4 at world 8 on one GPU: the C-ABI call (mi355_sw_batch_run_view), best and median of 20, score +
argmax and with traceback.  usage: c4_w8_time.py [world]   (environment: MI355_SW_POOL_THREADS / MI355_SW_POOL_SPIN_US for A/B)"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pgs = bench.load_package()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
out = []
for flags in (pgs.capi.SCORE_ONLY, 0):
    for _ in range(3):
        ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
    ts = []
    for _ in range(20):
        ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
        ts.append(ctx.last_call_s * 1e3)
    out.append("%s best %.3f median %.3f ms" % ("score+argmax" if flags else "with traceback", min(ts), float(np.median(ts))))
print("world %d, %d sequences, threads=%s spin=%s: %s" % (world, len(mine), os.environ.get("MI355_SW_POOL_THREADS", "default"),
                                                        os.environ.get("MI355_SW_POOL_SPIN_US", "default"), "; ".join(out)))
ctx.close()
