"""FULL-SIZE parity against fixtures made by the REAL reference (tests/golden/fullsize.json, generated in the build container
by tests/golden/make_fullsize_golden.py from oracle/_ref/ref_driver = the unmodified src/aligner/*.cpp of the reference):

* configs[2] shape, 150 bp x 50 Mbp (SURVEY.md §8(d) cfg 3): 16 float-engine and 64 uint8-engine alignments — score, pos, argmax
  cell and both consensus strings of every one — 4 / 8 of them reads cut from inside the planted repeats of the bench's
  repeat-rich reference (interspersed families, microsatellites, poly-A: thousands of near-equal candidates);
* configs[3] shape (cfg 4): all 561 356 UniProt-shaped sequences against P02232, sha256 over (score, pos) in database order,
  the first 32 and every 10 007th result in the clear.

The inputs are regenerated here from the seeds the fixture records (synth is deterministic); nothing of the reference is read."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "fullsize.json")
KEYS = ("score", "pos", "end_x", "end_y", "cons_x", "cons_y")


@pytest.fixture(scope="module")
def fix():
    if not os.path.exists(FIX):
        pytest.fail("tests/golden/fullsize.json is missing (tests/golden/make_fullsize_golden.py makes it in the build container)")
    with open(FIX) as f:
        return json.load(f)


def _check(ctx, reads, exp, sem, what):
    ctx.batch_upload(reads)
    got = ctx.batch_run(semantics=sem)
    bad = []
    for k, (g, e) in enumerate(zip(got, exp)):
        for key in KEYS:
            if g[key] != e[key]:
                bad.append((what, k, key, g[key] if key not in ("cons_x", "cons_y") else len(g[key]), e[key] if key not in ("cons_x", "cons_y") else len(e[key])))
                break
    return bad


def test_config3_full_size_against_the_real_reference(pgs, fix):
    """150 bp x 50 Mbp, both engines, plain and repeat-rich reference: every field of every alignment equals what the reference's
    own SWAligner<Similarity_Matrix> / SWAligner<Similarity_Matrix_Skewed> returned for the same inputs."""
    c3 = fix["config3"]
    for need in ("f32_plain", "f32_repeats", "u8_plain", "u8_repeats"):
        assert need in c3, "fixture incomplete: %s missing (run tests/golden/make_fullsize_golden.py)" % need
    n = c3["ref_len"]
    bad = []
    ctx = pgs.Context(0)
    try:
        ref = pgs.synth.dna(c3["plain"]["seed"], n)
        # the reads the fixture holds are the ones the generator makes from these seeds (the fixture is self-consistent)
        reads, _ = pgs.synth.reads_from_ref(ref, c3["read_seed"], 4, c3["read_len"])
        assert [r.tobytes().decode() for r in reads] == c3["plain_reads"][:4]
        ctx.set_reference(ref)
        del ref
        bad += _check(ctx, c3["plain_reads"][:len(c3["f32_plain"])], c3["f32_plain"], pgs.F32, "f32 plain")
        bad += _check(ctx, c3["plain_reads"][:len(c3["u8_plain"])], c3["u8_plain"], pgs.U8SAT, "u8 plain")
        rp = {k: v for k, v in c3["repeats"].items() if k != "seed"}
        rep, _ = pgs.synth.dna_repeats(c3["repeats"]["seed"], n, **rp)
        ctx.set_reference(rep)
        del rep
        bad += _check(ctx, c3["repeat_reads"][:len(c3["f32_repeats"])], c3["f32_repeats"], pgs.F32, "f32 repeats")
        bad += _check(ctx, c3["repeat_reads"][:len(c3["u8_repeats"])], c3["u8_repeats"], pgs.U8SAT, "u8 repeats")
    finally:
        ctx.close()
    assert len(c3["f32_plain"]) + len(c3["f32_repeats"]) >= 16 and len(c3["u8_plain"]) + len(c3["u8_repeats"]) >= 64
    assert not bad, "%d of the full-size alignments differ from the real reference: %r" % (len(bad), bad[:6])


def test_config4_all_561356_alignments_against_the_real_reference(pgs, fix):
    """The UniProt-shaped batch at FULL size: sha256 over the (score, pos) of all 561 356 alignments in database order equals the
    digest of the reference's own loop (src/mpi_sw_solve_uniprot.cpp:95-138, SWAligner<Similarity_Matrix>(db_seq, query))."""
    c4 = fix.get("config4")
    assert c4, "fixture incomplete: config4 missing (run tests/golden/make_fullsize_golden.py)"
    nseq = c4["sequences"]
    lens = pgs.synth.lognormal_lengths(c4["seed"], nseq)
    res = pgs.synth.protein(c4["seed"], int(lens.sum()))
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ctx = pgs.Context(0)
    try:
        ctx.set_reference(pgs.synth.P02232)
        ctx.batch_upload_packed(res, offs)
        out = ctx.batch_run(semantics=pgs.F32, raw=True)
        score, pos = np.asarray(out["score"]), np.asarray(out["pos"])
    finally:
        ctx.close()
    for k, (s, p) in enumerate(c4["first"]):
        assert (float(score[k]), int(pos[k])) == (s, p), ("first results", k, float(score[k]), int(pos[k]), s, p)
    for j, (s, p) in enumerate(c4["every_10007th"]):
        k = j * 10007
        assert (float(score[k]), int(pos[k])) == (s, p), ("sampled results", k, float(score[k]), int(pos[k]), s, p)
    assert float(score.astype(np.float64).sum()) == c4["sum_score"] and int(pos.astype(np.int64).sum()) == c4["sum_pos"]
    h = hashlib.sha256()
    h.update("".join("%.9g %d\n" % (float(s), int(p)) for s, p in zip(score.tolist(), pos.tolist())).encode())
    assert h.hexdigest() == c4["sha256"]
