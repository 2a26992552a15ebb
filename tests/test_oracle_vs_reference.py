"""Live differential: C restatement vs the real reference (oracle/_ref/ref_driver).  Runs only
where the reference build exists (this container); skipped on the GPU box."""
import random

import numpy as np
import pytest

from oracle import refproc as rp

pytestmark = pytest.mark.skipif(not rp.available(), reason="oracle/_ref not built (no /root/reference)")


def _rs(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def test_random_differential(oracle):
    rng = random.Random(7)
    cases = []
    for t in range(600):
        m = rng.choice([1, 2, 7, 31, 32, 33, 64, 96, 127, 128, 150])
        n = rng.choice([1, 2, 7, 31, 32, 33, 64, 96, 127, 128, 150, 400])
        if t % 5 == 0:
            n = max(1, m + rng.choice([-1, 0, 1]))
        y = _rs(rng, n)
        x = y[:m] if (t % 3 == 0 and n >= m) else _rs(rng, m)
        x = x + "A" * (m - len(x))
        cases.append((x, y, t % 2, rng.choice([(3.0, -3.0, 2.0), (2.0, -1.0, 1.0), (7.0, -5.0, 3.0)])))

    def go(idx):
        try:
            outs = rp.run([rp.align_cmd(cases[i][0], cases[i][1], cases[i][2], *cases[i][3]) for i in idx])
        except RuntimeError:
            if len(idx) == 1:       # reference aborts only on an all-zero matrix
                x, y, sem, sc = cases[idx[0]]
                assert oracle.align(x, y, sem, *sc)["score"] == 0
                return
            h = len(idx) // 2
            go(idx[:h]); go(idx[h:])
            return
        for i, o in zip(idx, outs):
            x, y, sem, sc = cases[i]
            exp = rp.parse_align(o)
            got = oracle.align(x, y, sem, *sc)
            assert {k: got[k] for k in exp} == exp, (x, y, sem, sc)
    go(list(range(len(cases))))


def test_wrapped_triangle_comes_first_in_the_skewed_order(oracle):
    """Similarity_Matrix_Skewed stores the bottom-right triangle (i + j > |y|) in the first columns of its skewed layout, so its
    find_index_of_maximum meets a saturated cell there BEFORE an equal one anywhere else (similaritymatrix.cpp:291-299, :330-364):
    the same 150 bp element at columns 1000.. and at the very end of the second sequence — the real reference and the restatement
    both report the copy at the end; without it, the first copy.  (What the uint8 engine's first-candidates step must respect,
    DESIGN.md §3.4.)"""
    rng = random.Random(11)
    elem = _rs(rng, 150)
    n = 3000
    y = list(_rs(rng, n))
    y[1000:1150] = elem
    y[n - 150:] = elem
    y = "".join(y)
    y2 = y[:n - 150] + _rs(rng, 150)
    outs = rp.run([rp.align_cmd(elem, y, 1, 3.0, -3.0, 2.0), rp.align_cmd(elem, y2, 1, 3.0, -3.0, 2.0)])
    for o, yy, where in zip(outs, (y, y2), ("end", "first")):
        exp = rp.parse_align(o)
        got = oracle.align(elem, yy, 1, 3.0, -3.0, 2.0)
        assert {k: got[k] for k in exp} == exp, where
        assert got["score"] == 255.0
        assert (got["pos"] >= n - 150) == (where == "end"), (where, got["pos"])


def test_uint8_scorings_that_truncate_differential(oracle):
    """The uint8 engine truncates its scoring to integers (_saturate, similaritymatrix.cpp:376-384): a gap of 0.37 or 0.5 becomes 0, a
    mismatch of -0.9 becomes 0.  Real reference against the restatement on 120 random pairs under such scorings — the semantics
    test_uint8_gap_truncated_to_zero_and_too_long_for_lds relies on at 14 000 rows."""
    rng = random.Random(23)
    cases = []
    for t in range(120):
        m = rng.choice([1, 5, 31, 32, 33, 64, 100, 150])
        n = rng.choice([1, 7, 33, 64, 127, 150, 400])
        if t % 6 == 0:
            n = m
        y = _rs(rng, n)
        x = y[:m] if (t % 3 == 0 and n >= m) else _rs(rng, m)
        x = x + "C" * (m - len(x))
        cases.append((x, y, rng.choice([(1.1, -0.9, 0.37), (3.0, -3.0, 0.5), (7.9, -1.2, 0.99), (2.0, -0.5, 1.0)])))

    def go(idx):
        try:
            outs = rp.run([rp.align_cmd(cases[i][0], cases[i][1], 1, *cases[i][2]) for i in idx])
        except RuntimeError:
            if len(idx) == 1:       # reference aborts only on an all-zero matrix
                x, y, sc = cases[idx[0]]
                assert oracle.align(x, y, 1, *sc)["score"] == 0
                return
            h = len(idx) // 2
            go(idx[:h]); go(idx[h:])
            return
        for i, o in zip(idx, outs):
            x, y, sc = cases[i]
            exp = rp.parse_align(o)
            got = oracle.align(x, y, 1, *sc)
            assert {k: got[k] for k in exp} == exp, (x, y, sc)
    go(list(range(len(cases))))


def test_fractional_and_cheap_gap_scorings_differential(oracle):
    """Float engine under scorings whose float32 sums round (0.7 / -0.3 / 0.1, 1.1 / -0.9 / 0.37), dyadic fractions (3.5 / -2.25 / 1.5)
    and cheap gaps (7 / -7 / 1: long gapped alignments): the restatement follows dp_func's operation order (similaritymatrix.cpp:49-54)
    bit for bit — scores compared as the reference prints them, positions and consensus strings exactly."""
    rng = random.Random(31)
    cases = []
    for t in range(160):
        m = rng.choice([2, 7, 31, 33, 64, 100, 150])
        n = rng.choice([7, 33, 64, 127, 150, 400, 900])
        y = _rs(rng, n)
        if t % 3 == 0 and n > m + 10:
            o = rng.randrange(0, n - m)
            x = list(y[o:o + m])
            for _ in range(max(1, m // 12)):
                x[rng.randrange(len(x))] = rng.choice("ACGT")
            if m > 20:
                del x[m // 2:m // 2 + 3]
            x = "".join(x)
        else:
            x = _rs(rng, m)
        cases.append((x, y, rng.choice([(0.7, -0.3, 0.1), (1.1, -0.9, 0.37), (3.5, -2.25, 1.5), (7.0, -7.0, 1.0), (0.5, -0.25, 0.25)])))

    def go(idx):
        try:
            outs = rp.run([rp.align_cmd(cases[i][0], cases[i][1], 0, *cases[i][2]) for i in idx])
        except RuntimeError:
            if len(idx) == 1:
                x, y, sc = cases[idx[0]]
                assert oracle.align(x, y, 0, *sc)["score"] == 0
                return
            h = len(idx) // 2
            go(idx[:h]); go(idx[h:])
            return
        for i, o in zip(idx, outs):
            x, y, sc = cases[i]
            exp = rp.parse_align(o)
            got = oracle.align(x, y, 0, *sc)
            # (the driver prints nine significant digits: the same float32)
            assert np.float32(got["score"]) == np.float32(exp["score"]), (x, y, sc, got["score"], exp["score"])
            assert {k: got[k] for k in exp if k != "score"} == {k: exp[k] for k in exp if k != "score"}, (x, y, sc)
    go(list(range(len(cases))))


def test_database_sequence_first_short_query_second_differential(oracle):
    """The UniProt driver's shape (src/mpi_sw_solve_uniprot.cpp:120): a database sequence as FIRST argument (2 .. 900 residues), the
    short query as SECOND (<= 200), protein alphabet, float engine, default scoring — planted pieces of the query (ties between equal
    copies, walks with gaps) and unrelated sequences; what sw_wave_prof_kernel and its windows are compared with on the device."""
    aa = "ACDEFGHIKLMNPQRSTVWY"
    rng = random.Random(41)
    cases = []
    for t in range(200):
        n = rng.choice([1, 2, 16, 17, 100, 144, 160, 200])
        y = _rs(rng, n, aa)
        m = rng.choice([2, 15, 63, 64, 65, 128, 129, 300, 900])
        x = list(_rs(rng, m, aa))
        if t % 2 == 0 and n >= 8:
            a = rng.randrange(0, n - 4)
            piece = list(y[a:a + rng.randrange(4, n - a + 1)])
            for _ in range(len(piece) // 10):
                piece[rng.randrange(len(piece))] = rng.choice(aa)
            if len(piece) > 12 and t % 4 == 0:
                del piece[5:7]
            reps = 2 if t % 6 == 0 else 1
            for r in range(reps):
                if len(piece) <= len(x):
                    at = rng.randrange(0, len(x) - len(piece) + 1)
                    x[at:at + len(piece)] = piece
        cases.append(("".join(x), y))

    def go(idx):
        try:
            outs = rp.run([rp.align_cmd(cases[i][0], cases[i][1], 0) for i in idx])
        except RuntimeError:
            if len(idx) == 1:
                assert oracle.align(cases[idx[0]][0], cases[idx[0]][1], 0)["score"] == 0
                return
            h = len(idx) // 2
            go(idx[:h]); go(idx[h:])
            return
        for i, o in zip(idx, outs):
            exp = rp.parse_align(o)
            got = oracle.align(cases[i][0], cases[i][1], 0)
            assert {k: got[k] for k in exp} == exp, cases[i]
    go(list(range(len(cases))))
