#!/usr/bin/env python3
"""Randomised differential stress run of the many-small-alignments batch (sw_wave_prof_kernel, checkpointed windows, the
host-driven fallback of walks that leave their window; sw_wave_kernel where the scores are not dyadic): batches of x's against
one short second sequence y, every field of every alignment against the CPU oracle, for a time budget.
Usage: python tests/stress_small_batches.py [seconds] [seed]   (or run(seconds, seed) from tests/test_gpu_stress.py)"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pgs = g._load_package()
from oracle import binding as ob  # noqa: E402

rng = None          # set by run()
findings = []       # the mismatch lines of the last run() (tests/test_gpu_stress.py puts them into its assertion message)


def _found(msg, flush=True):
    findings.append(msg)
    print(msg, flush=True)

KEYS = ("score", "pos", "end_x", "end_y", "cons_x", "cons_y")
ALPH = [b"ACDEFGHIKLMNPQRSTVWY"] * 3 + [b"ACGT", b"AC", bytes(range(65, 65 + 26))]
SC = [(3.0, -3.0, 2.0)] * 4 + [(2.0, -1.0, 1.0), (5.0, -4.0, 3.0), (1.0, -1.0, 1.0), (10.0, -2.0, 4.0), (3.5, -2.25, 1.5),
                               (2.0, -1.0, 0.5), (0.5, -0.25, 0.25), (64.0, -48.0, 32.0), (0.7, -0.3, 0.1), (1.1, -0.9, 0.37)]


def rseq(n, alpha):
    a = np.frombuffer(alpha, dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes()


def mutate(s, alpha, rate):
    a = np.frombuffer(alpha, dtype=np.uint8)
    out = bytearray()
    for ch in bytearray(s):
        u = rng.random()
        if u < rate / 4:
            continue
        if u < rate / 2:
            out.append(int(a[rng.integers(0, len(a))]))
        out.append(int(a[rng.integers(0, len(a))]) if rng.random() < rate else ch)
    return bytes(out)


def run(budget=120.0, seed=4321):
    """The stress loop for `budget` seconds from `seed`; returns (alignments, mismatches, walks that left their window)."""
    global rng
    rng = np.random.default_rng(seed)
    del findings[:]
    ctx = pgs.Context(0)
    t0 = time.time()
    ncase = nbad = nleft = 0
    rounds = 0
    while time.time() - t0 < budget:
        alpha = ALPH[int(rng.integers(0, len(ALPH)))]
        ylen = int(rng.choice([1, 2, 15, 16, 17, 100, 144, 159, 160, 161, 300, 320, 321, 500, 512]))
        y = rseq(ylen, alpha)
        sc = SC[int(rng.integers(0, len(SC)))]
        xs = []
        for _ in range(int(rng.integers(200, 1200))):
            m = int(rng.choice([0, 1, 2, 30, 63, 64, 65, 100, 127, 128, 129, 200, 400, 700, 1500]))
            kind = rng.random()
            if kind < 0.5 or m < 4:
                x = rseq(m, alpha)
            else:
                # a (possibly diverged, possibly repeated) piece of y planted somewhere in x
                a = int(rng.integers(0, ylen))
                piece = y[a:a + int(rng.integers(1, ylen - a + 1))]
                piece = mutate(piece, alpha, float(rng.choice([0.0, 0.03, 0.1, 0.25])))
                if rng.random() < 0.2:
                    piece = piece * int(rng.integers(2, 4))
                x = bytearray(rseq(max(m, len(piece)), alpha))
                at = int(rng.integers(0, len(x) - len(piece) + 1))
                x[at:at + len(piece)] = piece
                x = bytes(x)
            xs.append(x)
        with ThreadPoolExecutor(8) as ex:
            exp = list(ex.map(lambda x: ob.align(x, y, 0, *sc), xs))
        # three ways to the same answers: result objects (mi355_sw_batch_run), the struct-of-arrays view with consensus strings
        # (mi355_sw_batch_run_view: the device list's id-ordered records fill it directly), the view without traceback
        mode = rounds % 3
        if mode == 0:
            got = ctx.align_batch(xs, y, semantics=0, match=sc[0], mismatch=sc[1], gap=sc[2])
        else:
            ctx.set_reference(y)
            ctx.batch_upload(xs)
            raw = ctx.batch_run(semantics=0, match=sc[0], mismatch=sc[1], gap=sc[2], raw=True, flags=pgs.capi.SCORE_ONLY if mode == 2 else 0)
            got = []
            for k in range(len(xs)):
                g1 = dict(score=float(raw["score"][k]), pos=int(raw["pos"][k]), end_x=int(raw["end_x"][k]), end_y=int(raw["end_y"][k]))
                if mode == 1:
                    g1["cons_x"], g1["cons_y"] = ctx.consensus(k)
                else:                                                  # (no traceback: pos and the strings are not made)
                    g1["pos"], g1["cons_x"], g1["cons_y"] = exp[k]["pos"], exp[k]["cons_x"], exp[k]["cons_y"]
                got.append(g1)
        nleft += ctx.last_counters()["left_window"]
        for k, (ge, e) in enumerate(zip(got, exp)):
            ncase += 1
            for key in KEYS:
                if ge[key] != e[key]:
                    nbad += 1
                    if nbad <= 5:
                        _found("MISMATCH (mode %d) |y|=%d |x|=%d scoring %r alphabet %d: %s got %r expected %r" % (mode, ylen, len(xs[k]), sc, len(alpha), key, ge[key], e[key]),
                              flush=True)
                    break
        rounds += 1
        if rounds % 10 == 0:
            print("%d batches, %d alignments, %d mismatches, %d walks left their window (%.0f s)" % (rounds, ncase, nbad, nleft, time.time() - t0), flush=True)
    print("done: %d batches, %d alignments, %d mismatches, %d walks left their window" % (rounds, ncase, nbad, nleft), flush=True)
    ctx.close()
    return ncase, nbad, nleft


if __name__ == "__main__":
    _r = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
    sys.exit(1 if _r[1] else 0)
