#!/usr/bin/env python3
"""Randomised differential stress run: the HIP path (C-ABI) against the CPU oracle for a time budget.
Covers single alignments, ragged batches and the split aligner over a wide range of shapes, alphabets and
scorings.  Usage: python tests/stress.py [seconds] [seed]   (or run(seconds, seed) from tests/test_gpu_stress.py)"""
import os
import sys
import time

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

ctx = None
KEYS = ("score", "pos", "end_x", "end_y", "cons_x", "cons_y")
ALPH = [b"ACGT", b"ACGT", b"ACGT", b"AC", b"ACGTN", b"ACDEFGHIKLMNPQRSTVWY", bytes(range(65, 65 + 26))]
SC = [(3.0, -3.0, 2.0)] * 5 + [(2.0, -1.0, 1.0), (5.0, -4.0, 3.0), (1.0, -1.0, 1.0), (10.0, -2.0, 4.0), (3.5, -2.25, 1.5),
                               (4.0, -6.0, 3.0), (1.0, -3.0, 5.0), (7.0, -7.0, 1.0), (255.0, -200.0, 100.0), (0.5, -0.25, 0.25),
                               (0.7, -0.3, 0.1), (1.1, -0.9, 0.37), (0.1, -0.3, 0.7)]       # non-dyadic: float32 adds round


def rseq(n, alpha):
    a = np.frombuffer(alpha, dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes()


def mutate(s, alpha, rate):
    b = bytearray(s)
    a = np.frombuffer(alpha, dtype=np.uint8)
    out = bytearray()
    for ch in b:
        u = rng.random()
        if u < rate / 4:
            continue
        if u < rate / 2:
            out.append(int(a[rng.integers(0, len(a))]))
        out.append(int(a[rng.integers(0, len(a))]) if rng.random() < rate else ch)
    return bytes(out) or bytes(b[:1])


def plant(ref, m, alpha):
    n = len(ref)
    if n > m + 2 and rng.random() < 0.7:
        o = int(rng.integers(0, n - m))
        return mutate(ref[o:o + m], alpha, float(rng.choice([0.0, 0.02, 0.08, 0.2])))[:max(1, m)]
    return rseq(m, alpha)


def bisect_batch(qs, ref, sem, sc, exps):
    """A mismatching batch again under the switches of DESIGN.md §8.1: which path holds the difference.  The case is saved."""
    out = os.path.join(ROOT, "gpurun_out", "stress_case_%d.npz" % int(time.time()))
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        np.savez_compressed(out, ref=np.frombuffer(ref, dtype=np.uint8), sem=sem, sc=np.array(sc),
                            lens=np.array([len(q) for q in qs]), qs=np.frombuffer(b"".join(qs), dtype=np.uint8))
        _found("case saved to %s" % out)
    except OSError as e:
        print("case not saved: %r" % (e,), flush=True)
    for var in ({}, {"no_long": 1}, {"no_sample": 1}, {"no_satflag": 1}, {"no_opt_margin": 1}, {"no_requery": 1}, {"no_f16": 1}, {"force_f32": 1},
                {"no_strip": 1}, {"no_wave": 1}, {"no_strip_groups": 1}, {"no_first": 1}, {"no_unsat": 1}):
        c = pgs.Context(0)
        try:
            for k, v in var.items():
                c.set_option(k, v)
            res = c.align_batch(qs, ref, semantics=sem, match=sc[0], mismatch=sc[1], gap=sc[2])
            bad = [(i, len(qs[i]), [k for k in KEYS if res[i][k] != exps[i][k]]) for i in range(len(qs)) if any(res[i][k] != exps[i][k] for k in KEYS)]
            print("  %-22s %s  kernel: %s  counters: %s" % (var, "OK" if not bad else "BAD %r" % (bad[:3],), c.last_kernel()["name"][:110], c.last_counters()), flush=True)
            for i, m, keys in bad[:1]:
                print("     got score %r pos %r end %r/%r; expected score %r pos %r end %r/%r" % (res[i]["score"], res[i]["pos"], res[i]["end_x"], res[i]["end_y"],
                      exps[i]["score"], exps[i]["pos"], exps[i]["end_x"], exps[i]["end_y"]), flush=True)
            # the offending queries alone
            if bad and not var:
                for i, m, keys in bad[:2]:
                    r1 = c.align(qs[i], ref, sem, *sc)
                    print("     alone: %s (kernel %s)" % ("OK" if all(r1[k] == exps[i][k] for k in KEYS) else "BAD score %r" % r1["score"], c.last_kernel()["name"][:90]), flush=True)
        except Exception as e:
            print("  %-22s raised %r" % (var, e), flush=True)
        finally:
            c.close()


def run(budget=120.0, seed=12345):
    """The stress loop for `budget` seconds from `seed`; returns (cases, mismatches).  Every mismatch is printed with its
    case; a mismatching ragged batch is saved under gpurun_out/ and replayed under the switches of DESIGN.md."""
    global rng, ctx
    rng = np.random.default_rng(seed)
    del findings[:]
    ctx = pgs.Context(0)
    t0 = time.time()
    ncase = nbad = 0
    tick = t0
    while time.time() - t0 < budget:
        if time.time() - tick > 45:
            tick = time.time()
            print("... %d cases, %d mismatches, %.0f s" % (ncase, nbad, tick - t0), flush=True)
        alpha = ALPH[int(rng.integers(0, len(ALPH)))]
        kind = int(rng.integers(0, 13))
        ma, mi, gp = SC[int(rng.integers(0, len(SC)))]
        sem = int(rng.integers(0, 2))
        if kind <= 5:       # single alignment, oracle cost bounded to ~3e8 cells
            m = int(rng.choice([1, 5, 33, 40, 50, 64, 75, 100, 150, 152, 153, 250, 400, 512, 513, 1000, 2048, 2049, 3000, 6000,
                                11000, 17000]))
            nmax = max(2, int(3e8 // max(m, 1)))
            n = int(min(nmax, rng.choice([3, 150, 900, 1024, 5000, 40000, 300000, 2000000])))
            ref = rseq(n, alpha)
            q = plant(ref, m, alpha)
            lut = None
            if sem == 0 and rng.random() < 0.25:      # table scoring (float engine only): integer or fractional table
                lut = pgs.synth.make_lut(int(rng.integers(1, 1 << 30)), float(rng.choice([1.0, 1.0, 0.5, 1.25])))
                gp = float(rng.choice([1.0, 2.0, 3.0, 1.5]))
            exp = ob.align(q, ref, sem, ma, mi, gp, lut)
            try:
                got = ctx.align(q, ref, sem, ma, mi, gp, lut)
            except Exception as e:                   # an error is a finding too: report the case and go on
                got = {k: repr(e)[:80] for k in KEYS}
            bad = [k for k in KEYS if got[k] != exp[k]]
            if bad:
                nbad += 1
                _found("MISMATCH single m=%d n=%d sem=%d sc=%s lut=%s alpha=%d keys=%s got=%s exp=%s" %
                      (len(q), n, sem, (ma, mi, gp), lut is not None, len(alpha), bad, {k: got[k] for k in bad[:2]},
                       {k: exp[k] for k in bad[:2]}), flush=True)
            ncase += 1
        elif kind == 12:    # float engine beyond float16's exact range: the saturating sweep + re-evaluated flags
            n = int(rng.choice([20000, 60000, 120000]))
            ref = bytearray(rseq(n, alpha))
            base = int(rng.choice([690, 800, 1000, 1024, 1500, 2048]))
            qs = []
            for _ in range(int(rng.integers(2, 7))):
                m = max(1, base - int(rng.integers(0, 8)))
                q = plant(bytes(ref), m, alpha)
                qs.append(q)
                if rng.random() < 0.4 and len(q) + 10 < n:     # the same hit again somewhere else (ties / better copies)
                    at = int(rng.integers(0, n - len(q)))
                    ref[at:at + len(q)] = q
            if rng.random() < 0.15:
                c = alpha[:1]
                at = int(rng.integers(0, n // 2))
                ref[at:at + n // 3] = c * (n // 3)             # a background that saturates in a wide region
                qs[0] = c * len(qs[0])
            ref = bytes(ref)
            res = ctx.align_batch(qs, ref, semantics=0, match=ma, mismatch=mi, gap=gp)
            for q, got in zip(qs, res):
                exp = ob.align(q, ref, 0, ma, mi, gp)
                bad = [k for k in KEYS if got[k] != exp[k]]
                if bad:
                    nbad += 1
                    _found("MISMATCH long batch |q|=%d n=%d sc=%s keys=%s got=%s exp=%s" % (len(q), n, (ma, mi, gp), bad,
                          {k: got[k] for k in bad[:2]}, {k: exp[k] for k in bad[:2]}), flush=True)
                ncase += 1
        elif kind <= 7:     # ragged batch
            n = int(rng.choice([1500, 20000, 150000]))
            ref = rseq(n, alpha)
            qs = [plant(ref, int(rng.choice([0, 1, 20, 100, 150, 151, 300, 600, 1100, 2300])), alpha) if rng.random() > 0.05 else b""
                  for _ in range(int(rng.integers(1, 24)))]
            res = ctx.align_batch(qs, ref, semantics=sem, match=ma, mismatch=mi, gap=gp)
            exps = [ob.align(q, ref, sem, ma, mi, gp) for q in qs]
            anybad = False
            for q, got, exp in zip(qs, res, exps):
                bad = [k for k in KEYS if got[k] != exp[k]]
                if bad:
                    nbad += 1
                    anybad = True
                    _found("MISMATCH batch |q|=%d n=%d sem=%d sc=%s keys=%s (batch lengths %r)" % (len(q), n, sem, (ma, mi, gp), bad, [len(x) for x in qs]), flush=True)
                ncase += 1
            if anybad:
                bisect_batch(qs, ref, sem, (ma, mi, gp), exps)
        elif kind >= 10:    # many small whole problems against a short reference (device-built job lists, struct-of-arrays view)
            n = int(rng.choice([1, 40, 144, 300, 700, 1023]))
            ref = rseq(n, alpha)
            qs = [plant(ref, int(rng.choice([0, 1, 2, 17, 60, 144, 145, 300, 511, 512, 513, 900, 3000])), alpha) for _ in range(int(rng.integers(20, 400)))]
            ctx.set_reference(ref)
            ctx.batch_upload(qs)
            raw = ctx.batch_run(semantics=sem, match=ma, mismatch=mi, gap=gp, raw=True)
            for k, q in enumerate(qs):
                exp = ob.align(q, ref, sem, ma, mi, gp)
                cx, cy = ctx.consensus(k)
                got = dict(score=float(raw["score"][k]), pos=int(raw["pos"][k]), end_x=int(raw["end_x"][k]), end_y=int(raw["end_y"][k]),
                           cons_x=cx, cons_y=cy)
                bad = [key for key in KEYS if got[key] != exp[key]]
                if bad:
                    nbad += 1
                    _found("MISMATCH small |q|=%d n=%d sem=%d sc=%s keys=%s got=%s exp=%s" % (len(q), n, sem, (ma, mi, gp), bad,
                                                                                            {key: got[key] for key in bad[:2]}, {key: exp[key] for key in bad[:2]}), flush=True)
                ncase += 1
        else:               # split aligner
            n = int(rng.choice([3000, 30000, 120000]))
            m = int(rng.choice([20, 150, 400, 1200]))
            ref = rseq(n, alpha)
            q = plant(ref, m, alpha)
            npiece = int(rng.choice([1, 2, 3, 7, 17]))
            ratio = float(rng.choice([2.0, 1.0, 1.5]))
            sm, la = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            if ob.make_string_range(npiece, len(q), n, ratio) is None:
                continue
            exp = ob.align_split(q, ref, npiece, ratio, sm, la, ma, mi, gp)
            got = ctx.align_split(q, ref, npiece, ratio, sm, la, ma, mi, gp)
            bad = [k for k in ("score", "pos", "cons_x", "cons_y", "piece") if got[k] != exp[k]]
            if bad:
                nbad += 1
                _found("MISMATCH split m=%d n=%d npiece=%d sm=%d la=%d sc=%s keys=%s" % (len(q), n, npiece, sm, la, (ma, mi, gp), bad), flush=True)
            ncase += 1
    print("stress: %d cases in %.0f s, %d mismatches (seed %d)" % (ncase, time.time() - t0, nbad, seed), flush=True)
    ctx.close()
    return ncase, nbad


if __name__ == "__main__":
    _, _nbad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    sys.exit(1 if _nbad else 0)
