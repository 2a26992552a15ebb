#!/usr/bin/env python3
"""Config 4 (BASELINE.json): UniProt-shaped many-alignment batch — N protein sequences (log-normal lengths,
median ~290, mean ~360) as FIRST argument, the 144-aa P02232 query as SECOND (src/mpi_sw_solve_uniprot.cpp:120),
default identity scoring 3/-3, gap 2, float32 engine.  Prints timing and the batch best (score, index)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pgs = g._load_package()
P02232 = pgs.synth.P02232
n = int(sys.argv[1]) if len(sys.argv) > 1 else 561_356
lens = pgs.synth.lognormal_lengths(5, n)
tot = int(lens.sum())
allres = pgs.synth.protein(5, tot)
offs = np.concatenate([[0], np.cumsum(lens)])
seqs = [allres[offs[k]:offs[k + 1]].tobytes() for k in range(n)]
cells = float(tot) * len(P02232)
print("%d sequences, mean len %.1f, max %d, %.3g cells" % (n, lens.mean(), lens.max(), cells), flush=True)
ctx = pgs.Context(0)
ctx.set_reference(P02232)
t0 = time.time()
ctx.batch_upload(seqs)
t1 = time.time()
# each mode twice: the first call of a mode also sizes (and pins) its staging buffers
for flags, name in ((pgs.capi.SCORE_ONLY, "score+argmax (first call)"), (pgs.capi.SCORE_ONLY, "score+argmax"),
                    (0, "full (traceback) (first call)"), (0, "full (traceback)")):
    t2 = time.time()
    out = ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
    t3 = time.time()
    tm = ctx.last_timings()
    best = int(out["score"].argmax())
    print(name, json.dumps(dict(wall_s=t3 - t2, device_ms=tm["total_us"] / 1e3, gcups_wall=cells / (t3 - t2) * 1e-9,
                                best_score=float(out["score"][best]), best_index=best, upload_s=t1 - t0)), flush=True)
ctx.close()
