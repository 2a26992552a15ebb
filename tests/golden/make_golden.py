#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/ref_driver, built by
oracle/build_ref.sh from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

The fixtures are data: inputs + the reference's outputs.  Committed so that the CPU restatement
(oracle/sw_oracle.c) and the HIP path can be checked where /root/reference does not exist.
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refproc as rp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")


def rs(rng, n, alpha):
    return "".join(rng.choice(alpha) for _ in range(n))


def mutate(rng, s, alpha, sub=0.04, indel=0.02):
    out = []
    for ch in s:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            out.append(rng.choice(alpha))
        out.append(rng.choice(alpha) if rng.random() < sub else ch)
    return "".join(out) or s[:1]


def run_align_cases(cases):
    """cases: list of dict(x,y,sem,match,mismatch,gap) -> fills 'expect'.  The reference aborts
    on an all-zero matrix, so such cases are dropped by bisection."""
    def go(idx):
        try:
            outs = rp.run([rp.align_cmd(c["x"], c["y"], c["sem"], c["match"], c["mismatch"], c["gap"])
                           for c in (cases[i] for i in idx)])
            for i, o in zip(idx, outs):
                cases[i]["expect"] = rp.parse_align(o)
        except RuntimeError:
            if len(idx) > 1:
                h = len(idx) // 2
                go(idx[:h]); go(idx[h:])
    go(list(range(len(cases))))
    return [c for c in cases if "expect" in c]


def main():
    assert rp.available(), "build oracle/_ref first (make -C oracle)"
    rng = random.Random(20261003)
    out = {}

    # 1. known answers held by the reference's own tests
    kat = [
        dict(name="wikipedia_u8 test_localaligner.cpp:24-27,53-58", x="GGTTGACTA", y="TGTTACGG", sem=1, match=3.0, mismatch=-3.0, gap=2.0),
        dict(name="wikipedia_f32", x="GGTTGACTA", y="TGTTACGG", sem=0, match=3.0, mismatch=-3.0, gap=2.0),
        dict(name="two_hits_f32 SURVEY App.B", x="ACGTACGTTG", y="TTTTACGTACGTTGCCCCCCACGTACGTTGGGG", sem=0, match=3.0, mismatch=-3.0, gap=2.0),
        dict(name="two_hits_u8 SURVEY App.B", x="ACGTACGTTG", y="TTTTACGTACGTTGCCCCCCACGTACGTTGGGG", sem=1, match=3.0, mismatch=-3.0, gap=2.0),
        dict(name="x_longer_f32", x="TTTTACGTACGTTGCCCCCC", y="ACGTACGTTG", sem=0, match=3.0, mismatch=-3.0, gap=2.0),
        dict(name="x_longer_u8", x="TTTTACGTACGTTGCCCCCC", y="ACGTACGTTG", sem=1, match=3.0, mismatch=-3.0, gap=2.0),
        dict(name="custom_2_-1_1_f32", x="GGTTGACTA", y="TGTTACGG", sem=0, match=2.0, mismatch=-1.0, gap=1.0),
        dict(name="custom_2_-1_1_u8", x="GGTTGACTA", y="TGTTACGG", sem=1, match=2.0, mismatch=-1.0, gap=1.0),
    ]
    out["kat"] = run_align_cases(kat)
    assert len(out["kat"]) == len(kat)

    # matrices (test_localaligner.cpp:30-50 commented matrix; test_skewedmatrix.cpp:39-66)
    mats = []
    for x, y in [("GGTTGACTA", "TGTTACGG"), ("GGTTGACTA", "TGTTACG"), ("TGTTACG", "GGTTGACTA"),
                 ("CGACCATAG", "TCGGCGGCC"), (rs(rng, 40, "ACGT"), rs(rng, 40, "ACGT")),
                 (rs(rng, 33, "AC"), rs(rng, 70, "AC")), (rs(rng, 70, "AC"), rs(rng, 33, "AC"))]:
        for sem in (0, 1):
            line = rp.run(["matrix %s 3.0 -3.0 2.0 %s %s" % (rp.SEM[sem], x, y)])[0]
            mats.append(dict(x=x, y=y, sem=sem, match=3.0, mismatch=-3.0, gap=2.0,
                             cells=[float(v) for v in line.split()]))
    out["matrix"] = mats

    # 2. randomised alignments, both semantics, shapes around the SIMD width (32), square,
    #    |x|>|y|, saturating (>85 matches), several scorings
    cases = []
    sizes = [1, 2, 3, 5, 8, 9, 16, 31, 32, 33, 40, 63, 64, 65, 90, 100, 125, 130, 150]
    scorings = [(3.0, -3.0, 2.0)] * 3 + [(2.0, -1.0, 1.0), (5.0, -4.0, 3.0), (1.0, -1.0, 1.0), (10.0, -2.0, 4.0),
                                         (3.5, -2.25, 1.5), (200.0, -100.0, 90.0)]
    for t in range(420):
        alpha = rng.choice(["ACGT", "ACGT", "ACGT", "AC", "ACGTN"])
        m = rng.choice(sizes)
        n = rng.choice(sizes + [200, 300, 700, 1500])
        if t % 7 == 0:
            n = m + rng.choice([-1, 0, 0, 1])
            n = max(1, n)
        y = rs(rng, n, alpha)
        if rng.random() < 0.6 and n > m + 2:
            o = rng.randrange(0, n - m)
            x = mutate(rng, y[o:o + m], alpha)
        else:
            x = rs(rng, m, alpha)
        ma, mi, g = rng.choice(scorings)
        cases.append(dict(x=x, y=y, sem=t % 2, match=ma, mismatch=mi, gap=g))
    # repeats / ties / plateaus
    for t in range(40):
        unit = rs(rng, rng.choice([1, 2, 3, 7]), "ACGT")
        y = (unit * 400)[:rng.choice([60, 200, 900])]
        x = (unit * 400)[:rng.choice([10, 40, 100, 150])]
        cases.append(dict(x=x, y=y, sem=t % 2, match=3.0, mismatch=-3.0, gap=2.0))
    out["align"] = run_align_cases(cases)

    # 3. table (LUT) scoring through std::function (smithwaterman.cpp:23-38), integer and
    #    fractional entries; u8 semantics only probes ('A','A') and ('A','T')
    lut_cases = []
    for t in range(60):
        alpha = rng.choice(["ACGT", "ACDEFGHIKLMNPQRSTVWY"])
        m = rng.choice([5, 20, 33, 64, 100, 144])
        n = rng.choice([10, 50, 100, 360, 800])
        y = rs(rng, n, alpha)
        x = mutate(rng, y[rng.randrange(0, max(1, n - m)):][:m], alpha) if rng.random() < 0.5 else rs(rng, m, alpha)
        lut_cases.append(dict(x=x, y=y, sem=0 if t % 4 else 1, seed=rng.randrange(1, 1 << 30),
                              scale=rng.choice([1.0, 1.0, 0.25, 2.0]), gap=rng.choice([1.0, 2.0, 0.5, 3.0])))
    good = []
    for c in lut_cases:
        try:
            o = rp.run([rp.alignlut_cmd(c["x"], c["y"], c["sem"], c["seed"], c["scale"], c["gap"])])[0]
            c["expect"] = rp.parse_align(o)
            good.append(c)
        except RuntimeError:
            pass
    out["alignlut"] = good

    # 4. OMPParallelLocalAligner, serial build (plocalaligner.cpp:105-143)
    split = []
    for t in range(80):
        m = rng.choice([10, 25, 60, 125])
        n = rng.choice([400, 1000, 2500, 4980])
        npiece = rng.choice([1, 2, 3, 4, 8, 17])
        ratio = rng.choice([2.0, 2.0, 1.0, 1.5])
        y = rs(rng, n, "ACGT")
        if t % 5 == 0:
            unit = y[:m]
            y = (y[:100] + unit + y[100:n // 2] + unit + y[n // 2:])[:n]   # two equal hits
            x = unit
        else:
            x = mutate(rng, y[rng.randrange(0, n - m):][:m], "ACGT")
        sm, la = rng.choice([(0, 0), (1, 1), (1, 0), (0, 1)])
        ma, mi, g = rng.choice([(3.0, -3.0, 2.0), (3.0, -3.0, 2.0), (2.0, -1.0, 1.0)])
        try:
            o = rp.run([rp.split_cmd(x, y, sm, la, npiece, ratio, ma, mi, g)])[0]
        except RuntimeError:
            continue   # reference assert (overlap > piecelen) or no match
        e = rp.parse_align(o)
        split.append(dict(x=x, y=y, sm=sm, la=la, npiece=npiece, ratio=ratio, match=ma, mismatch=mi, gap=g,
                          expect=dict(score=e["score"], pos=e["pos"], cons_x=e["cons_x"], cons_y=e["cons_y"])))
    out["split"] = split

    # 5. _make_string_range (plocalaligner.cpp:44-67)
    ranges = []
    for npiece, s, l, r in [(1, 10, 100, 2.0), (2, 10, 100, 2.0), (4, 10, 100, 2.0), (17, 125, 4980, 2.0),
                            (8, 125, 4980, 2.0), (16, 150, 1000000, 2.0), (64, 10000, 250000000, 2.0),
                            (3, 7, 1000, 1.5), (5, 33, 777, 0.5), (2, 150, 50000000, 2.0)]:
        o = rp.run(["range %d %d %d %r" % (npiece, s, l, r)])[0].split()
        ranges.append(dict(npiece=npiece, short=s, long=l, ratio=r,
                           ranges=[[int(o[2 * k]), int(o[2 * k + 1])] for k in range(npiece)]))
    out["range"] = ranges

    # 6. skewed index maps (test_skewedmatrix.cpp:5-37): every cell of several shapes
    maps = []
    for m, n in [(9, 7), (7, 9), (9, 9), (1, 5), (5, 1), (40, 33), (33, 40)]:
        cmds = ["true2raw %d %d %d %d" % (m, n, ti, tj) for ti in range(n + 1) for tj in range(m + 1)]
        o = rp.run(cmds)
        maps.append(dict(m=m, n=n, raw=[[int(v) for v in line.split()] for line in o]))
    out["true2raw"] = maps

    with open(os.path.join(HERE, "ref_cases.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print({k: len(v) for k, v in out.items()})

    # 7. config 1 (sw_solve_small): data/data_small as shipped -> per-read expected outputs for the
    #    first 48 reads and sha256 digests of the full 1170-read output (SURVEY App. B format).
    fa = "".join(open(os.path.join(REF, "data/data_small/genome.chr22.5K.fa")).read().split("\n")[1:])
    reads = []
    with open(os.path.join(REF, "data/data_small_ground_truth.csv")) as f:
        next(f)
        for line in f:
            t = line.rstrip("\n").split(",")
            reads.append((int(t[0]), t[2], int(t[3])))
    small = dict(ref=fa, reads=[r[1] for r in reads], sam_pos=[r[2] for r in reads], digests={}, first={})
    for sem in (0, 1):
        outs = rp.run([rp.align_cmd(r[1], fa, sem) for r in reads], timeout=3600)
        lines = []
        first = []
        for (idx, seq, _), o in zip(reads, outs):
            e = rp.parse_align(o)
            lines.append("%d,%g,%d,%s,%s\n" % (idx, e["score"], e["pos"], e["cons_x"], e["cons_y"]))
            if idx < 48:
                first.append(e)
        small["digests"][rp.SEM[sem]] = hashlib.sha256("".join(lines).encode()).hexdigest()
        small["first"][rp.SEM[sem]] = first
        print(rp.SEM[sem], small["digests"][rp.SEM[sem]])
    # split variants of SURVEY App. B
    for name, sm, la, npiece in [("f32_npiece8", 0, 0, 8), ("u8_npiece17", 1, 1, 17)]:
        outs = rp.run([rp.split_cmd(r[1], fa, sm, la, npiece, 2.0) for r in reads], timeout=3600)
        lines = []
        for (idx, seq, _), o in zip(reads, outs):
            e = rp.parse_align(o)
            lines.append("%d,%g,%d,%s,%s\n" % (idx, e["score"], e["pos"], e["cons_x"], e["cons_y"]))
        small["digests"][name] = hashlib.sha256("".join(lines).encode()).hexdigest()
        print(name, small["digests"][name])
    with open(os.path.join(HERE, "data_small.json"), "w") as f:
        json.dump(small, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
