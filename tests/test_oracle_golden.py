"""The CPU restatement (oracle/sw_oracle.c) against fixtures produced by the REAL reference
(tests/golden/make_golden.py -> oracle/_ref/ref_driver) and the reference's own gtest known
answers.  CPU only."""
import hashlib

import numpy as np
import pytest


def _cmp(got, exp, what):
    for k in ("score", "pos", "end_x", "end_y", "cons_x", "cons_y"):
        if k in exp and exp[k] != -1:
            assert got[k] == exp[k], "%s: %s differs: got %r expected %r" % (what, k, got[k], exp[k])


def test_gtest_known_answers(oracle):
    # test/test_localaligner.cpp:24-27 and :53-58
    r = oracle.align("GGTTGACTA", "TGTTACGG", oracle.U8SAT)
    assert r["score"] == 13 and r["pos"] == 2
    assert r["cons_x"] == "CAGTTG" and r["cons_y"] == "CA-TTG"


def test_commented_matrix(oracle):
    # test/test_localaligner.cpp:33-42 (rows = "GGTTGACTA", cols = "TGTTACGG")
    exp = np.array([[0, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 3, 1, 0, 0, 0, 3, 3], [0, 0, 3, 1, 0, 0, 0, 3, 6],
                    [0, 3, 1, 6, 4, 2, 0, 1, 4], [0, 3, 1, 4, 9, 7, 5, 3, 2], [0, 1, 6, 4, 7, 6, 4, 8, 6],
                    [0, 0, 4, 3, 5, 10, 8, 6, 5], [0, 0, 2, 1, 3, 8, 13, 11, 9], [0, 3, 1, 5, 4, 6, 11, 10, 8],
                    [0, 1, 0, 3, 2, 7, 9, 8, 7]], dtype=np.float32)
    assert np.array_equal(oracle.fill("GGTTGACTA", "TGTTACGG", oracle.F32), exp)
    assert np.array_equal(oracle.fill("GGTTGACTA", "TGTTACGG", oracle.U8SAT).astype(np.float32), exp)


def test_skewed_equals_normal(oracle):
    # test/test_skewedmatrix.cpp:39-66
    for x, y in (("GGTTGACTA", "TGTTACG"), ("TGTTACG", "GGTTGACTA")):
        assert np.array_equal(oracle.fill(x, y, oracle.F32), oracle.fill(x, y, oracle.U8SAT).astype(np.float32))


def test_index_maps_roundtrip(oracle):
    # test/test_skewedmatrix.cpp:5-37
    for m, n in ((9, 7), (7, 9), (9, 9)):
        for ti in range(n + 1):
            for tj in range(m + 1):
                ri, rj = oracle.true2raw(m, n, ti, tj)
                assert oracle.raw2true(m, n, ri, rj) == (ti, tj)


def test_kat(oracle, golden):
    for c in golden["kat"]:
        _cmp(oracle.align(c["x"], c["y"], c["sem"], c["match"], c["mismatch"], c["gap"]), c["expect"], c["name"])


def test_matrices(oracle, golden):
    for c in golden["matrix"]:
        H = oracle.fill(c["x"], c["y"], c["sem"], c["match"], c["mismatch"], c["gap"]).astype(np.float32)
        exp = np.array(c["cells"], dtype=np.float32).reshape(len(c["x"]) + 1, len(c["y"]) + 1)
        assert np.array_equal(H, exp), (c["x"], c["y"], c["sem"])


def test_align_cases(oracle, golden):
    assert len(golden["align"]) > 400
    for k, c in enumerate(golden["align"]):
        _cmp(oracle.align(c["x"], c["y"], c["sem"], c["match"], c["mismatch"], c["gap"]), c["expect"], "align[%d]" % k)


def test_alignlut_cases(oracle, golden, pgs):
    for k, c in enumerate(golden["alignlut"]):
        lut = pgs.synth.make_lut(c["seed"], c["scale"])
        _cmp(oracle.align(c["x"], c["y"], c["sem"], gap=c["gap"], lut=lut), c["expect"], "alignlut[%d]" % k)


def test_split_cases(oracle, golden):
    assert len(golden["split"]) > 50
    for k, c in enumerate(golden["split"]):
        got = oracle.align_split(c["x"], c["y"], c["npiece"], c["ratio"], c["sm"], c["la"], c["match"],
                                 c["mismatch"], c["gap"])
        _cmp(got, c["expect"], "split[%d]" % k)


def test_string_ranges(oracle, golden):
    for c in golden["range"]:
        got = oracle.make_string_range(c["npiece"], c["short"], c["long"], c["ratio"])
        assert got == [tuple(r) for r in c["ranges"]], c


def test_true2raw(oracle, golden):
    for c in golden["true2raw"]:
        m, n = c["m"], c["n"]
        k = 0
        for ti in range(n + 1):
            for tj in range(m + 1):
                assert list(oracle.true2raw(m, n, ti, tj)) == c["raw"][k]
                k += 1


def test_data_small_first_reads(oracle, data_small):
    ref = data_small["ref"]
    for sem, name in ((oracle.F32, "f32"), (oracle.U8SAT, "u8")):
        for k, exp in enumerate(data_small["first"][name]):
            _cmp(oracle.align(data_small["reads"][k], ref, sem), exp, "data_small[%s][%d]" % (name, k))


@pytest.mark.parametrize("sem,name", [(0, "f32"), (1, "u8")])
def test_data_small_digest(oracle, data_small, sem, name):
    """Config 1 (sw_solve_small) at full size: sha256 over all 1170 reads equals the digest of the
    real reference's output (== SURVEY.md Appendix B)."""
    ref = data_small["ref"]
    lines = []
    for k, read in enumerate(data_small["reads"]):
        r = oracle.align(read, ref, sem)
        lines.append("%d,%g,%d,%s,%s\n" % (k, r["score"], r["pos"], r["cons_x"], r["cons_y"]))
    assert hashlib.sha256("".join(lines).encode()).hexdigest() == data_small["digests"][name]
    survey = {"f32": "7e18fbb7c0e5dda4bbc8a9840c9a207fa58cf1958e260365f21dc19e56b13fde",
              "u8": "9dd2dc41d44fd90c120e27950a1f8e5a263b5ac93de6349836d43efa74169d03"}
    assert data_small["digests"][name] == survey[name]


def test_data_small_split_digest(oracle, data_small):
    ref = data_small["ref"]
    lines = []
    for k, read in enumerate(data_small["reads"]):
        r = oracle.align_split(read, ref, 17, 2.0, oracle.U8SAT, oracle.U8SAT)
        lines.append("%d,%g,%d,%s,%s\n" % (k, r["score"], r["pos"], r["cons_x"], r["cons_y"]))
    assert hashlib.sha256("".join(lines).encode()).hexdigest() == data_small["digests"]["u8_npiece17"]
    assert data_small["digests"]["u8_npiece17"] == "15243408f24069bab5c479a40d99e93468a3a91197047d51b2b88e9f6c43539c"


def test_no_match_is_defined(oracle):
    # deliberate divergence from the reference's UB
    r = oracle.align("AAAA", "CCCCCC", oracle.F32)
    assert r["score"] == 0 and r["pos"] == 0 and r["cons_x"] == "" and r["cons_y"] == ""


def test_lean_locate_equals_full_argmax(oracle, golden):
    """sw_oracle_locate (no matrix) against the matrix-based argmax on every golden alignment case."""
    for c in golden["align"][:200] + golden["kat"]:
        if c["expect"]["score"] <= 0:
            continue
        got = oracle.locate(c["x"], c["y"], c["sem"], c["match"], c["mismatch"], c["gap"])
        assert got == (c["expect"]["score"], c["expect"]["end_x"], c["expect"]["end_y"]), c


def test_trace_from_equals_reference_traceback(oracle, golden):
    """sw_oracle_trace_from (traceback from a given start cell; the companion of the lean locate for full-size checks)
    against the real reference's consensus / pos on every golden alignment: started at the reference's argmax it must
    reproduce the reference's walk."""
    n = 0
    for c in golden["align"] + golden["kat"]:
        e = c["expect"]
        if e["score"] <= 0:
            continue
        w = oracle.trace_from(c["x"], c["y"], c["sem"], e["end_x"], e["end_y"], c["match"], c["mismatch"], c["gap"])
        assert (w["score"], w["cons_x"], w["cons_y"], w["pos"]) == (e["score"], e["cons_x"], e["cons_y"], e["pos"]), c
        n += 1
    assert n > 400


def test_eval_mismatch_counts_of_data_small(data_small):
    """SURVEY.md §0.5 / py/eval.py:102-121 on the committed golden positions: the reference's float engine differs from the
    SAM POS column on 188 of 1170 reads, its uint8 engine on 222 (the oracle reproduces every position, so its count is
    the reference's)."""
    from oracle import binding as ob
    ref = data_small["ref"].encode()
    for sem, want in ((ob.F32, 188), (ob.U8SAT, 222)):
        bad = sum(1 for q, p in zip(data_small["reads"], data_small["sam_pos"]) if ob.align(q.encode(), ref, sem)["pos"] != p)
        assert bad == want, (sem, bad)


def test_stress_fixture_is_what_its_script_makes(tmp_path):
    """tests/golden/stress_seed777_long_query_in_batch.npz (the batch tests/stress.py seed 777 found, replayed on the device by
    test_lone_long_query_inside_a_batch) is reproduced byte for byte by the committed script, and the oracle's answer for its long
    query is the one DESIGN.md §3.7 quotes (score 10362, an alignment that spans 5963 columns)."""
    import os
    import subprocess
    import sys
    import numpy as np
    from oracle import binding as ob
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    subprocess.check_call([sys.executable, os.path.join(here, "make_stress_case.py"), str(tmp_path)], stdout=subprocess.DEVNULL)
    made = np.load(str(tmp_path / "stress_seed777_case2.npz"))
    kept = np.load(os.path.join(here, "stress_seed777_long_query_in_batch.npz"))
    assert sorted(made.files) == sorted(kept.files)
    for k in kept.files:
        assert np.array_equal(made[k], kept[k]), k
    lens = kept["lens"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    q = kept["qs"].tobytes()[offs[6]:offs[7]]
    r = ob.align(q, kept["ref"].tobytes(), 0, *[float(v) for v in kept["sc"]])
    assert (len(q), r["score"], r["pos"], r["end_y"]) == (2300, 10362.0, 11966, 17929), r


def test_fullsize_fixture_is_complete_and_pins_the_oracle(pgs):
    """tests/golden/fullsize.json (made by the REAL reference at 150 bp x 50 Mbp and over all 561 356 UniProt-shaped alignments,
    tests/golden/make_fullsize_golden.py) holds what tests/test_gpu_fullsize.py needs, and the C restatement agrees with it at
    full size: the lean rolling-column locate + the window traceback of one float-engine and one uint8-engine alignment give
    the reference's score, argmax cell, pos and consensus strings."""
    import json
    import os
    from oracle import binding as ob
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize.json")
    fix = json.load(open(path))
    c3, c4 = fix["config3"], fix["config4"]
    assert len(c3["f32_plain"]) + len(c3["f32_repeats"]) >= 16 and len(c3["u8_plain"]) + len(c3["u8_repeats"]) >= 64
    assert len(c3["repeat_reads"]) >= 8 and c4["sequences"] == 561356 and len(c4["sha256"]) == 64
    ref = pgs.synth.dna(c3["plain"]["seed"], c3["ref_len"])
    reads, _ = pgs.synth.reads_from_ref(ref, c3["read_seed"], 3, c3["read_len"])
    assert [r.tobytes().decode() for r in reads] == c3["plain_reads"][:3]          # the fixture's reads are what the seeds make
    refb = ref.tobytes()
    for sem, key in ((ob.F32, "f32_plain"), (ob.U8SAT, "u8_plain")):
        q, e = c3["plain_reads"][2].encode(), c3[key][2]
        mx, ix, iy = ob.locate(q, refb, sem)
        assert (mx, ix, iy) == (e["score"], e["end_x"], e["end_y"]), (key, mx, ix, iy, e["score"], e["end_x"], e["end_y"])
        lo = max(0, iy - 20_000)
        w = ob.trace_from(q, refb[lo:iy], sem, ix, iy - lo)
        assert (w["cons_x"], w["cons_y"], w["pos"] + lo) == (e["cons_x"], e["cons_y"], e["pos"]), key
