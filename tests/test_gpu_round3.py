"""GPU parity, round 3: every A/B switch of the library lights an alternate kernel instance or pipeline that is also a
production fallback — each must give the oracle's answers on the fuzz case list; the pipelined long-query score kernel
(sw_long_kernel); the reference-sharding finish call (mi355_sw_align_scored_range)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cmp(got, exp, what):
    assert got["score"] == exp["score"], "%s: score %r != %r" % (what, got["score"], exp["score"])
    for k in ("pos", "end_x", "end_y", "cons_x", "cons_y"):
        if k in exp and exp[k] != -1:
            assert got[k] == exp[k], "%s: %s differs: got %r expected %r" % (what, k, got[k], exp[k])


@pytest.fixture(scope="module")
def fuzz_cases(pgs, oracle):
    """The case list of test_fuzz_paths_vs_oracle (same seed, same construction) plus batches that reach the instances
    single alignments do not (two queries per register, whole-wavefront tiles, strips), with the oracle's answers."""
    rng = np.random.default_rng(20261003)
    lens = [1, 2, 15, 16, 17, 63, 64, 65, 100, 104, 105, 127, 128, 129, 150, 152, 153, 160, 200, 208, 209, 255, 256, 257,
            300, 511, 512, 513, 600]
    scorings = [(3.0, -3.0, 2.0)] * 4 + [(2.0, -1.0, 1.0), (5.0, -4.0, 3.0), (10.0, -2.0, 1.0), (1.0, -1.0, 4.0),
                                         (3.5, -2.25, 1.5), (2.0, -7.0, 2.0), (100.0, -90.0, 60.0)]
    singles = []
    for t in range(220):
        m = int(rng.choice(lens))
        n = int(rng.choice([1, 7, 150, 151, 152, 600, 1023, 1024, 1025, 3000, 9000, 40000]))
        kind = t % 5
        if kind == 0:
            unit = pgs.synth.dna(int(rng.integers(1, 1 << 30)), int(rng.integers(1, 9))).tobytes()
            ref = (unit * (n // len(unit) + 1))[:n]
        else:
            ref = pgs.synth.dna(int(rng.integers(1, 1 << 30)), n).tobytes()
        if kind in (1, 2) and n > m + 2:
            o = int(rng.integers(0, n - m))
            q = bytearray(ref[o:o + m])
            for i in range(m):
                if rng.random() < 0.04:
                    q[i] = b"ACGT"[int(rng.integers(0, 4))]
            q = bytes(q)
        elif kind == 0:
            q = (ref * 3)[:m] if len(ref) >= 1 else b"A" * m
            q = (q * (m // max(1, len(q)) + 1))[:m]
        else:
            q = pgs.synth.dna(int(rng.integers(1, 1 << 30)), m).tobytes()
        sem = int(rng.integers(0, 2))
        sc = scorings[int(rng.integers(0, len(scorings)))]
        singles.append((q, ref, sem, sc, oracle.align(q, ref, sem, *sc)))
    batches = []
    ref = pgs.synth.dna(5151, 70_000)
    refb = ref.tobytes()
    qs = [pgs.synth.read_from_ref(ref, 6000 + k, m, sub_rate=0.03, indel_rate=0.005)[0].tobytes()
          for k, m in enumerate((40, 100, 150, 150, 152, 300, 300, 600, 640, 1000, 1000, 2048, 2500, 2600, 700, 125))]
    qs += [b"", pgs.synth.dna(99, 150).tobytes()]
    for sem in (0, 1):
        for sc in ((3.0, -3.0, 2.0), (3.5, -3.25, 2.0)):
            batches.append((qs, refb, sem, sc, [oracle.align(q, refb, sem, *sc) for q in qs]))
    lone = []
    for k, m in enumerate((700, 1500, 2100, 3000)):
        q = pgs.synth.read_from_ref(ref, 6100 + k, m, sub_rate=0.02, indel_rate=0.004)[0].tobytes()
        for sem in (0, 1):
            lone.append((q, refb, sem, (3.0, -3.0, 2.0), oracle.align(q, refb, sem)))
    return singles, batches, lone


# every switch that selects another kernel instance / pipeline for the same answer
SWITCHES = ["no_f16", "no_unsat", "no_sample", "no_satflag", "no_solo", "no_wave", "no_comb", "no_twin", "no_wide",
            "no_strip", "no_quant", "no_f16_wide", "no_devlist", "no_strip_groups", "u8_long_twin", "long_twin", "no_long",
            "no_requery", "slot=16", "strip_r=24", "few_r=5", "long_pipes=2"]


def test_option_names_cover_the_switch_list(pgs):
    names = set(pgs.capi.option_names())
    for sw in SWITCHES:
        assert sw.split("=")[0] in names, sw
    c = pgs.Context(0)
    try:
        with pytest.raises(pgs.MI355Error):
            c.set_option("no_such_switch")
    finally:
        c.close()


@pytest.mark.parametrize("switch", ["default"] + SWITCHES)
def test_every_switch_gives_the_oracle_answers(pgs, fuzz_cases, switch):
    """The alternates behind the switches are the library's FALLBACK targets too (a table that does not fit float16, a gap
    beyond 2040, a flag budget that overflows ...): each one runs the whole fuzz list, bit-exact against the oracle."""
    singles, batches, lone = fuzz_cases
    c = pgs.Context(0)
    try:
        if switch != "default":
            k, _, v = switch.partition("=")
            c.set_option(k, v or True)
        bad = []
        for t, (q, ref, sem, sc, exp) in enumerate(singles + lone):
            got = c.align(q, ref, sem, *sc)
            for key in ("score", "pos", "end_x", "end_y", "cons_x", "cons_y"):
                if got[key] != exp[key]:
                    bad.append((t, len(q), len(ref), sem, sc, key, got[key], exp[key]))
                    break
        for qs, refb, sem, sc, exps in batches:
            for q, got, exp in zip(qs, c.align_batch(qs, refb, semantics=sem, match=sc[0], mismatch=sc[1], gap=sc[2]), exps):
                for key in ("score", "pos", "end_x", "end_y", "cons_x", "cons_y"):
                    if got[key] != exp[key]:
                        bad.append(("batch", len(q), sem, sc, key, got[key], exp[key]))
                        break
        assert not bad, "%s: %d mismatches, first %r" % (switch, len(bad), bad[:3])
    finally:
        c.close()


@pytest.mark.parametrize("sem", [0, 1])
def test_long_kernel_vs_oracle(pgs, oracle, sem):
    """sw_long_kernel (lone query beyond 2048 rows, strips pipelined over a workgroup): lengths around the strip shapes
    (R = 20 / 24, one to nine strips), hits at the very start / end of the reference and across tile borders, an unrelated
    query (background maximum with ties), references with N; one and two tiles per workgroup; short tiles (many tiles,
    warm-up in front of each).  Every result bit-exact against the oracle; the kernel that ran is checked by name."""
    ref = pgs.synth.dna(7001, 150_000)
    refb = bytearray(ref.tobytes())
    refb[5000:5020] = b"N" * 20
    refb = bytes(refb)
    c = pgs.Context(0)
    try:
        for chunk in (None, 8192):
            c.set_option("chunk", chunk)
            for pipes in (None, 2):
                c.set_option("long_pipes", pipes)
                for k, m in enumerate((2049, 2560, 2561, 3072, 5000, 7681, 10_000, 12_288)):
                    if pipes == 2 and m > 64 * 24 * 8:
                        continue
                    o = [0, 150_000 - m, 70_000, 3000][k % 4]
                    q = bytearray(refb[o:o + m])
                    rng = np.random.default_rng(100 + k)
                    for i in rng.choice(m, m // 50, replace=False):
                        q[i] = b"ACGT"[int(rng.integers(0, 4))]
                    q = bytes(q).replace(b"N", b"A")
                    exp = oracle.align(q, refb, sem)
                    got = c.align(q, refb, sem)
                    _cmp(got, exp, "long kernel sem=%d m=%d chunk=%r pipes=%r" % (sem, m, chunk, pipes))
                    if m <= 10_240:                                                 # (12 288 rows x 6 codes: profile beyond the LDS)
                        assert "sw_long_kernel" in c.last_kernel()["name"], c.last_kernel()["name"]
                q = pgs.synth.dna(7100 + sem, 4000).tobytes()                       # unrelated: background maximum
                _cmp(c.align(q, refb, sem), oracle.align(q, refb, sem), "long kernel, unrelated query sem=%d" % sem)
    finally:
        c.close()


def test_long_kernel_other_scorings(pgs, oracle):
    """Integer scorings other than the default on sw_long_kernel (larger match: other scale; cheap gap: longer warm-up),
    and the fallbacks around it: fractional scoring and a table (float16 profile not exact) take the strip-mined instance."""
    ref = pgs.synth.dna(7201, 90_000)
    refb = ref.tobytes()
    q = pgs.synth.read_from_ref(ref, 7202, 3000, sub_rate=0.03, indel_rate=0.005)[0].tobytes()
    c = pgs.Context(0)
    try:
        for sc, want_long in (((5.0, -4.0, 3.0), True), ((2.0, -1.0, 1.0), True), ((100.0, -90.0, 60.0), True),
                              ((3.5, -2.25, 1.5), False)):
            _cmp(c.align(q, refb, 0, *sc), oracle.align(q, refb, 0, *sc), "long kernel scoring %r" % (sc,))
            assert ("sw_long_kernel" in c.last_kernel()["name"]) == want_long, (sc, c.last_kernel()["name"])
    finally:
        c.close()


def test_align_scored_range_is_the_pieces_alignment(pgs, oracle):
    """mi355_sw_score_ranges + mi355_sw_align_scored_range (reference sharding: the owner of the winning piece finishes it
    from the sweep's keys): equal to aligning the piece as a stand-alone problem (plocalaligner.cpp:132-137), both engines,
    short and long queries, other scoring for the finish (re-sweep), and stale keys are refused."""
    ref = pgs.synth.dna(7301, 300_000)
    refb = ref.tobytes()
    c = pgs.Context(0)
    try:
        c.set_reference(refb)
        for m in (150, 3000):
            q = pgs.synth.read_from_ref(ref, 7302 + m, m, sub_rate=0.02, indel_rate=0.004)[0].tobytes()
            c.batch_upload([q])
            for sem in (0, 1):
                for npiece in (2, 5):
                    ranges = pgs.capi.make_string_range(npiece, m, len(refb), 2.0)
                    mx = c.score_ranges(ranges, semantics=sem)[:, 0]
                    for k, (lo, hi) in enumerate(ranges):
                        exp = oracle.align(q, refb[lo:hi], sem)
                        assert mx[k] == exp["score"], (m, sem, npiece, k)
                        _cmp(c.align_scored_range(k, semantics=sem), exp, "scored range m=%d sem=%d piece %d/%d" % (m, sem, k, npiece))
                    # another scoring for the finish: swept again under it
                    exp = oracle.align(q, refb[ranges[1][0]:ranges[1][1]], sem, 2.0, -1.0, 1.0)
                    _cmp(c.align_scored_range(1, semantics=sem, match=2.0, mismatch=-1.0, gap=1.0), exp, "scored range, other scoring")
        c.batch_upload([q])                                            # a new batch: the old keys must not be used
        with pytest.raises(pgs.MI355Error):
            c.align_scored_range(0, semantics=0)
    finally:
        c.close()
