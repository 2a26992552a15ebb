#!/usr/bin/env python3
"""A saved batch (npz: ref, lens, qs, sc [, sem]) again on the device under the switches of DESIGN.md §8.1, every result against the
oracle: which path holds a difference.  Usage: python tests/replay_case.py case.npz [...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pgs = g._load_package()
from oracle import binding as ob  # noqa: E402

KEYS = ("score", "pos", "end_x", "end_y", "cons_x", "cons_y")
VARIANTS = ({}, {"no_long": 1}, {"no_sample": 1}, {"no_satflag": 1}, {"no_opt_margin": 1}, {"no_requery": 1}, {"no_f16": 1}, {"force_f32": 1},
            {"no_strip": 1}, {"no_wave": 1}, {"no_strip_groups": 1}, {"no_first": 1}, {"no_unsat": 1}, {"long_pipes": 1}, {"long_pipes": 2},
            {"long_pipes": 4}, {"no_long_p32": 1}, {"long_groups": 2}, {"chunk": 4096}, {"chunk": 8192}, {"chunk": 32768}, {"long_r": 24}, {"long_wgs": 4},
            {"no_sample": 1, "no_opt_margin": 1}, {"trace": 1})
rc = 0
for path in sys.argv[1:]:
    d = np.load(path)
    ref = d["ref"].tobytes()
    lens = d["lens"]
    allq = d["qs"].tobytes()
    offs = np.concatenate([[0], np.cumsum(lens)])
    qs = [allq[offs[k]:offs[k + 1]] for k in range(len(lens))]
    sc = tuple(float(v) for v in d["sc"])
    sem = int(d["sem"]) if "sem" in d else 0
    exps = [ob.align(q, ref, sem, *sc) for q in qs]
    print("%s: %d queries %r vs %d columns, engine %d, scoring %r" % (os.path.basename(path), len(qs), list(map(int, lens)), len(ref), sem, sc), flush=True)
    for var in VARIANTS:
        c = pgs.Context(0)
        try:
            for k, v in var.items():
                c.set_option(k, v)
            res = c.align_batch(qs, ref, semantics=sem, match=sc[0], mismatch=sc[1], gap=sc[2])
            bad = [(i, len(qs[i]), [k for k in KEYS if res[i][k] != exps[i][k]]) for i in range(len(qs)) if any(res[i][k] != exps[i][k] for k in KEYS)]
            print("  %-22s %s  kernel: %s  counters: %s" % (var, "OK" if not bad else "BAD %r" % (bad[:3],), c.last_kernel()["name"][:100], c.last_counters()), flush=True)
            if bad:
                rc = 1
                i = bad[0][0]
                print("     got score %r pos %r end %r/%r; expected score %r pos %r end %r/%r" % (res[i]["score"], res[i]["pos"], res[i]["end_x"], res[i]["end_y"],
                      exps[i]["score"], exps[i]["pos"], exps[i]["end_x"], exps[i]["end_y"]), flush=True)
                if not var:
                    for i, m, keys in bad[:2]:
                        r1 = c.align(qs[i], ref, sem, *sc)
                        print("     query %d alone: %s (kernel %s)" % (i, "OK" if all(r1[k] == exps[i][k] for k in KEYS) else "BAD score %r" % r1["score"], c.last_kernel()["name"][:90]),
                              flush=True)
        except Exception as e:
            print("  %-22s raised %r" % (var, e), flush=True)
        finally:
            c.close()
sys.exit(rc)
