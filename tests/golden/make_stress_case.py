#!/usr/bin/env python3
"""How tests/golden/stress_seed777_long_query_in_batch.npz was made: tests/stress.py's generator replayed on the CPU with seed 777
(device and oracle replaced by stubs that consume no random numbers), saving the ragged batches with a 2300 bp query against 20 000
columns at 7 / -7 / 1; the second one is the batch that came out wrong on the device.
Usage: python tests/golden/make_stress_case.py [outdir]   (writes stress_seed777_case<k>.npz there; default: next to this file)"""
import sys, os, types, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as real_ob
OUTDIR = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
class Stop(Exception): pass
R = dict(score=0.0,pos=0,end_x=0,end_y=0,cons_x="",cons_y="",piece=0)
class FakeOb:
    @staticmethod
    def align(*a, **k): return R
    @staticmethod
    def align_split(*a, **k): return R
    make_string_range = staticmethod(real_ob.make_string_range)
found = []
class FakeCtx:
    def __init__(self, *a): pass
    def align(self, *a, **k): return R
    def align_split(self, *a, **k): return R
    def align_batch(self, qs, ref, **kw):
        if len(ref) == 20000 and kw.get("match") == 7.0 and kw.get("gap") == 1.0 and kw.get("semantics") == 0 and any(len(q) == 2300 for q in qs):
            found.append(1)
            np.savez_compressed(os.path.join(OUTDIR, "stress_seed777_case%d.npz" % len(found)), ref=np.frombuffer(ref, dtype=np.uint8),
                                lens=np.array([len(q) for q in qs]), qs=np.frombuffer(b"".join(qs), dtype=np.uint8), sc=np.array([kw["match"], kw["mismatch"], kw["gap"]]))
            print("saved case", len(found), [len(q) for q in qs], flush=True)
            if len(found) >= 3: raise Stop()
        return [R for _ in qs]
    def set_reference(self, *a): pass
    def batch_upload(self, qs): self.n = len(qs)
    def batch_run(self, **kw):
        n = self.n
        return dict(score=np.zeros(n), pos=np.zeros(n, int), end_x=np.zeros(n, int), end_y=np.zeros(n, int))
    def consensus(self, k): return "", ""
    def close(self): pass
    def last_kernel(self): return {"name": ""}
    def last_counters(self): return {}
src = open(os.path.join(ROOT, "tests", "stress.py")).read()
src = src.replace("pgs = g._load_package()", "pgs = g._load_package(); pgs.Context = FakeCtx")
src = src.replace("from oracle import binding as ob  # noqa: E402", "ob = FakeOb")
sys.argv = ["stress.py", "100000", "777"]
g = {"__name__": "__main__", "__file__": os.path.join(ROOT, "tests", "stress.py"), "FakeCtx": FakeCtx, "FakeOb": FakeOb}
try:
    exec(compile(src, "stress.py", "exec"), g)
except Stop:
    print("stopped after", g.get("ncase"), "cases")
