#!/usr/bin/env python3
"""Config 5 (BASELINE.json): one 10 kbp query vs a 250 Mbp reference, whole-reference alignment and
OMPParallelLocalAligner-style split (overlap = 2x query).  Prints timings; self-consistency only at this
size (the CPU oracle cannot hold a 10k x 250M matrix): split and whole-reference results must agree."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pgs = g._load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
npiece = int(sys.argv[3]) if len(sys.argv) > 3 else 16
t0 = time.time()
ref = pgs.synth.dna(6, n)
q, off = pgs.synth.read_from_ref(ref, 7, m, sub_rate=0.01, indel_rate=0.001)
print("generated in %.1fs, planted at %d" % (time.time() - t0, off), flush=True)
ctx = pgs.Context(0)
out = {}
for sem, name in ((0, "f32"), (1, "u8")):
    t0 = time.time()
    ctx.set_reference(ref)
    ctx.batch_upload([q.tobytes()])
    t1 = time.time()
    r = ctx.batch_run(semantics=sem)[0]
    t2 = time.time()
    tm = ctx.last_timings()
    cells = float(m) * n
    out[name] = dict(score=r["score"], pos=r["pos"], end_y=r["end_y"], cons_len=len(r["cons_x"]), upload_s=t1 - t0,
                     align_s=t2 - t1, score_kernel_ms=tm["score_us"] / 1e3, locate_ms=tm["locate_us"] / 1e3,
                     trace_ms=tm["trace_us"] / 1e3, kernel_gcups=cells / tm["score_us"] * 1e-3, e2e_gcups=cells / (t2 - t1) * 1e-9)
    print(name, json.dumps(out[name]), flush=True)
    t0 = time.time()
    ranges = pgs.capi.make_string_range(npiece, m, n, 2.0)
    mx = ctx.score_ranges(ranges, semantics=sem)
    tm = ctx.last_timings()
    t1 = time.time()
    print(name, "split %d pieces: maxima %s winner %d, score kernel %.1f ms, wall %.2fs" %
          (npiece, mx[:, 0].astype(int).tolist(), int(mx[:, 0].argmax()), tm["score_us"] / 1e3, t1 - t0), flush=True)
    assert mx[:, 0].max() == r["score"]
ctx.close()
