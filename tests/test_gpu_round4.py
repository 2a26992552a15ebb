"""GPU parity, round 4: that every A/B switch ENGAGES (mi355_sw_last_path changes as the switch says, results unchanged),
the finish of a lone long query from the state its sweep saved, the retry on an expired wait between workgroups, device-derived
launch sizing, the split aligner on the winner-only sweep."""
import re

import numpy as np
import pytest

from switch_inputs import build, run_input

pytestmark = pytest.mark.gpu

KEYS = ("score", "pos", "end_x", "end_y", "cons_x", "cons_y")


def _cmp(got, exp, what):
    for k in KEYS:
        assert got[k] == exp[k], "%s: %s differs: got %r expected %r" % (what, k, got[k], exp[k])


def _set(c, switch, on=True):
    for part in switch.split("+"):
        k, _, v = part.partition("=")
        c.set_option(k, (v or True) if on else None)


# switch -> (input of tests/switch_inputs.py that reaches its alternate, base switches that stay on in both runs,
#            regex the path must match WITH the switch, regex it must match WITHOUT it — and not the other way round)
ENGAGE = {
    "no_f16": ("batch150_f32", "", r"score\[cell=i16", r"score\[cell=f16"),
    "no_unsat": ("batch150_u8", "", r"score\[cell=u8f16[^\]]*unsat=0", r"score\[cell=f16[^\]]*unsat=1"),
    "no_sample": ("batch150_f32", "", r"score\[[^\]]*sampled=0", r"score\[[^\]]*sampled=1"),
    "u8_sample_short": ("batch1000_u8", "no_u8_early", r"score\[[^\]]*sampled=0", r"score\[[^\]]*sampled=1"),
    "no_u8_early": ("batch1000_u8", "", r"score\[", r"^(?!.*score\[).*u8_early"),
    "no_satflag": ("batch1000_f32", "", r"score\[cell=i16[^\]]*satflag=0", r"score\[cell=f16[^\]]*satflag=1"),
    "no_solo": ("single150_f32", "", r"strip\[|wave\[", r"solo\["),
    "no_wave": ("uniprot_shape", "", r"^exact\[[^ ]*$", r"wave\["),
    "no_comb": ("single150_f32", "", r"twin=1,comb=0", r"twin=1,comb=1"),
    "no_twin": ("single150_f32", "", r"score\[cell=f32[^\]]*twin=0", r"score\[cell=f16[^\]]*twin=1"),
    "no_wide": ("batch600_f32", "", r"score\[[^\]]*SL=16,R=32,strips=1", r"score\[[^\]]*SL=64"),
    "no_strip": ("single1000_f32", "", r"exact\[", r"strip\["),
    "no_quant": ("single400_long_ref", "", r"score\[[^\]]*pow2=1", r"score\[[^\]]*pow2=0"),
    "no_devlist": ("uniprot_shape", "", r"^(?!.*devlist)", r"devlist\["),
    "no_strip_groups": ("single1000_f32", "", r"mode=dirs,grouped=0", r"mode=dirs,grouped=1"),
    "u8_long_twin": ("single1000_u8", "no_u8_early", r"score\[cell=f16[^\]]*twin=1", r"score\[cell=f32[^\]]*twin=0"),
    "long_twin": ("single1000_f32", "", r"score\[cell=i16[^\]]*twin=1", r"score\[cell=f32[^\]]*twin=0"),
    "no_long": ("single3000_f32", "", r"score\[cell=f32[^\]]*strips=1", r"long\["),
    "strip_r=24": ("single5000_f32", "no_long", r"score\[[^\]]*R=24,strips=1", r"score\[[^\]]*R=20,strips=1"),
    "no_requery": ("polya_f32", "", r"whole_again", r"requery"),
    "slot=16": ("batch150_f32", "", r"score\[[^\]]*SL=16,R=10", r"score\[[^\]]*SL=8,R=19"),
    "few_r=5": ("single1000_f32", "", r"strip\[R=5", r"strip\[R=3"),
    "long_pipes=2": ("single3000_f32", "", r"long\[[^\]]*pipes=2", r"long\[[^\]]*pipes=8"),
    "no_long_p32": ("single3000_f32", "", r"long\[[^\]]*p32=0", r"long\[[^\]]*p32=1"),
    "long_groups=2": ("single3000_f32", "", r"long\[[^\]]*groups=2", r"long\[[^\]]*groups=1"),
    "force_f32": ("batch150_f32", "", r"score\[cell=f32", r"score\[cell=f16"),
    "no_opt_margin": ("single3000_f32", "", r"long\[[^\]]*opt_margin=0", r"long\[[^\]]*opt_margin=1"),
    "no_wave_prof": ("uniprot_shape", "", r"devlist\[[^\]]*prof=0", r"devlist\[[^\]]*prof=1"),
    "no_wave_window": ("uniprot_shape", "", r"devlist\[[^\]]*windows=0", r"devlist\[[^\]]*windows=1"),
    "no_first": ("polya_u8", "", r"requery", r"first_settled"),
    "no_wave_pieces": ("uniprot_shape", "", r"devlist\[[^\]]*pieces=0", r"devlist\[[^\]]*pieces=1"),
    "no_wave_f16": ("uniprot_shape", "", r"devlist\[[^\]]*f16=0", r"devlist\[[^\]]*f16=1"),
    "no_devlist_by_id": ("uniprot_shape", "", r"devlist_results\[by_id=0", r"devlist_results\[by_id=1"),
    "no_long_save": ("single3000_f32", "", r"long\[[^\]]*saved=0[^ ]* strip\[[^\]]*mode=max", r"long\[[^\]]*saved=1[^ ]* saved_locate"),
}

_expected = {}


def _oracle_results(pgs, oracle, name):
    """The oracle's answers for input `name` (None where the reference is too long for it: those compare against the default path)."""
    if name not in _expected:
        kind, qs, ref, sem = build(pgs, name)
        if len(ref) > 2_000_000:
            _expected[name] = None
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(8) as ex:
                _expected[name] = list(ex.map(lambda q: oracle.align(q, ref, sem), qs))
    return _expected[name]


def test_engagement_table_covers_every_switch():
    """Every switch of the result-preserving list (tests/test_gpu_round3.py SWITCHES) has an engagement case here — a switch
    without a reachable input gets deleted, not listed (round 4: no_f16_wide went that way)."""
    from test_gpu_round3 import SWITCHES
    missing = [s for s in SWITCHES if s not in ENGAGE and s != "assume_cus=32"]    # (assume_cus: test_assumed_cu_count_resizes_the_launches)
    assert not missing, missing


@pytest.mark.parametrize("switch", sorted(ENGAGE))
def test_switch_engages(pgs, oracle, switch):
    """The switch changes WHICH kernels / pipeline ran — as recorded by the library itself (mi355_sw_last_path) — and not the
    results: with it the path matches the alternate's tag and not the default's, without it the other way round."""
    name, base, want_on, want_off = ENGAGE[switch]
    exp = _oracle_results(pgs, oracle, name)
    c = pgs.Context(0)
    try:
        if base:
            _set(c, base)
        path_off, res_off = run_input(pgs, c, name, want_results=True)
        _set(c, switch)
        path_on, res_on = run_input(pgs, c, name, want_results=True)
    finally:
        c.close()
    on, off = " ".join(path_on), " ".join(path_off)
    assert re.search(want_on, on), "%s on %s: the alternate did not run: %s" % (switch, name, on)
    assert re.search(want_off, off), "%s on %s: the default is not what the table says: %s" % (switch, name, off)
    assert not re.search(want_on, off) or want_on.startswith("^(?!"), "%s on %s: the default already matches the alternate's tag: %s" % (switch, name, off)
    assert not re.search(want_off, on), "%s on %s: the default's kernel still ran: %s" % (switch, name, on)
    for k, (a, b) in enumerate(zip(res_on, res_off)):
        _cmp(a, b, "%s on %s, alignment %d: switch vs default" % (switch, name, k))
    if exp is not None:
        for k, (a, e) in enumerate(zip(res_on, exp)):
            _cmp(a, e, "%s on %s, alignment %d vs oracle" % (switch, name, k))


def test_assumed_cu_count_resizes_the_launches(pgs, oracle):
    """Launch sizing comes from the device (hipGetDeviceProperties; option assume_cus overrides the CU count): a context that
    believes in 32 CUs picks other tile lengths and never more waiting workgroups than that many CUs hold — same answers."""
    kind, qs, ref, sem = build(pgs, "single3000_f32")
    exp = oracle.align(qs[0], ref, sem)
    c = pgs.Context(0)
    try:
        _cmp(c.align(qs[0], ref, sem), exp, "default sizing")
        c.set_option("assume_cus", 32)
        _cmp(c.align(qs[0], ref, sem), exp, "assume_cus=32")
        c.set_option("assume_cus", None)
        kind, qs, ref, sem = build(pgs, "batch150_f32")
        a = c.align_batch(qs, ref, semantics=sem)
        k_default = c.last_kernel()
        c.set_option("assume_cus", 8)
        b = c.align_batch(qs, ref, semantics=sem)
        k_small = c.last_kernel()
        assert k_small["chunk_len"] > k_default["chunk_len"], (k_default["chunk_len"], k_small["chunk_len"])   # fewer, longer tiles
        for k, (x, y) in enumerate(zip(a, b)):
            _cmp(y, x, "assume_cus=8, read %d" % k)
    finally:
        c.close()


def test_expired_wait_between_workgroups_is_retried_on_the_non_waiting_layout(pgs, oracle):
    """sw_long_kernel with the strips of a tile dealt to several workgroups waits across workgroups; where a device does not
    hold them all at once the wait expires.  The call then sweeps again on a layout whose waits stay inside one workgroup
    instead of failing (option fault_inject = long_stall makes every such wait expire at once)."""
    ref = pgs.synth.dna(9901, 400_000)
    q = pgs.synth.read_from_ref(ref, 9902, 10_000, sub_rate=0.02, indel_rate=0.003)[0].tobytes()    # 8 strips: two workgroups per tile
    refb = ref.tobytes()
    exp = oracle.align(q, refb, 0)
    c = pgs.Context(0)
    try:
        _cmp(c.align(q, refb, 0), exp, "long query")
        assert re.search(r"long\[[^\]]*groups=2", " ".join(c.last_path())), c.last_path()
        assert c.last_counters()["wait_retries"] == 0
        c.set_option("fault_inject", "long_stall")
        _cmp(c.align(q, refb, 0), exp, "long query, waits between workgroups expire")
        path = " ".join(c.last_path())
        assert c.last_counters()["wait_retries"] == 1, c.last_counters()
        assert re.search(r"long\[[^\]]*groups=1", path) or "score[" in path, path
        c.set_option("fault_inject", None)
        _cmp(c.align(q, refb, 0), exp, "long query again")                                           # the context stays usable
        assert c.last_counters()["wait_retries"] == 0
    finally:
        c.close()


def test_finish_from_saved_state(pgs, oracle):
    """A lone long query's locate and traceback start from the columns and strip rows its sweep saved (host_saved.h): blocks
    with known left column and top row instead of windows behind a zero border.  Hits at the start, in the middle and at the end
    of the reference, across tile borders, with indels (the walk leaves the diagonal), a reference with N, the full and the
    optimistic warm-up margin, the split aligner and best_range + align_scored_range: bit-exact against the oracle, the counters
    say which finish ran, and the zero-border path (no_long_save) gives the same."""
    n = 600_000
    ref = bytearray(pgs.synth.dna(9911, n).tobytes())
    ref[250_000:250_030] = b"N" * 30
    refb = bytes(ref)
    arr = np.frombuffer(refb, dtype=np.uint8)
    qs = []
    for k, (o, m) in enumerate(((0, 4000), (n - 5000, 5000), (200_000, 6000), (300_000 - 3000, 7000), (100_000, 2500))):
        q = pgs.synth.read_from_ref(arr[o:o + m + 64], 9920 + k, m, sub_rate=0.02, indel_rate=0.004)[0].tobytes()
        qs.append(q.replace(b"N", b"A"))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        exps = list(ex.map(lambda q: oracle.align(q, refb, 0), qs))
    c = pgs.Context(0)
    try:
        used = 0
        for var in ({}, {"no_opt_margin": 1}, {"long_sub": 1024}, {"no_long_save": 1}):
            for k, v in var.items():
                c.set_option(k, v)
            for q, exp in zip(qs, exps):
                _cmp(c.align(q, refb, 0), exp, "saved finish %r m=%d" % (var, len(q)))
                cnt = c.last_counters()
                if "no_long_save" in var:
                    assert cnt["saved_locates"] == cnt["saved_traces"] == 0 and "saved=0" in " ".join(c.last_path())
                else:
                    assert cnt["saved_locates"] >= 1 or cnt["saved_fallbacks"] >= 1, (var, cnt, c.last_path())
                    used += cnt["saved_traces"]
            for k in var:
                c.set_option(k, None)
        assert used >= 5, used                                                    # the traceback from saved state did run (full margin: always)
        # through the split aligner (winner-only sweep, the winner finished from its keys) and best_range + align_scored_range
        q, m = qs[2], len(qs[2])
        got = c.align_split(q, refb, 5, 2.0, 0, 0)
        exp = oracle.align_split(q, refb, 5, 2.0, 0, 0)
        for k in ("score", "pos", "cons_x", "cons_y", "piece"):
            assert got[k] == exp[k], ("align_split", k, got[k], exp[k])
        assert c.last_timings()["score_launches"] == 1, c.last_timings()        # ONE sweep: no second one for the winner
        c.set_reference(refb)
        c.batch_upload([q])
        ranges = pgs.capi.make_string_range(5, m, n, 2.0)
        best, which, _ = c.best_range(ranges)
        w = int(which[0])
        r = c.align_scored_range(w)
        lo, hi = ranges[w]
        _cmp(r, oracle.align(q, refb[lo:hi], 0), "align_scored_range from saved state")
        cnt = c.last_counters()
        assert cnt["saved_traces"] + cnt["saved_fallbacks"] >= 1, cnt
    finally:
        c.close()


def test_float16_first_pass_boundaries(pgs, oracle):
    """The many-small-alignments batch on packed float16 cells (sw_wave_prof16_kernel, DESIGN.md lemma L13) at the edges of what
    it may decide: scores just below / at / above 128 units (the key's four free mantissa bits), equal maxima in neighbouring
    columns of one lane and in different lanes (first column wins, then first row), problems paired with a much longer / an
    empty neighbour, an odd number of problems, scorings whose unit is not 1, and the scorings / stream lengths the kernel
    must refuse (match * (|y| + 1) >= 2048 units; streams beyond 65 000 rows when pieces are off)."""
    y = pgs.synth.P02232.encode()                                               # 144 letters: nine columns per lane
    rnd = lambda seed, n: pgs.synth.protein(seed, n).tobytes()
    xs = []
    for k, run in enumerate((41, 42, 43, 44, 60, 143)):                         # 3 * run: 123, 126, 129, 132, 180, 429
        xs.append(rnd(100 + k, 200)[:77] + y[50:50 + run if 50 + run <= 144 else 144][:run] + rnd(200 + k, 90))
    xs.append(rnd(300, 30) + y[10:30] + rnd(301, 45) + y[10:30] + rnd(302, 20))           # the same 20 letters twice: equal maxima, first row wins
    xs.append(rnd(303, 25) + y[12:30] + rnd(304, 31) + y[72:90] + rnd(305, 20))           # equal maxima in different lanes: first column wins
    xs.append(y[100:120] + rnd(306, 300) + y[20:40])                                      # ... the later row holds the earlier column
    xs += [b"", y[:1], rnd(307, 5000), b"", rnd(308, 33), y, y[::-1], rnd(309, 1601), rnd(310, 1599)]
    xs += [rnd(400 + k, 64 * (k % 7) + k) for k in range(41)]                             # odd count, lengths around the segments
    c = pgs.Context(0)
    try:
        for sc, f16 in (((3.0, -3.0, 2.0), True), ((0.5, -0.25, 0.25), True), ((64.0, -48.0, 32.0), True),
                        ((14.0, -3.0, 2.0), True),       # 14 * 145 = 2030 < 2048 units: the last scoring the cells hold exactly
                        ((15.0, -3.0, 2.0), False),      # 15 * 145 = 2175: refused
                        ((3.0, -2049.0, 2.0), False)):   # a mismatch score beyond the cell's range: refused
            exp = [oracle.align(x, y, 0, *sc) for x in xs]
            for flags in (0, pgs.capi.SCORE_ONLY):
                got = c.align_batch(xs, y, semantics=0, match=sc[0], mismatch=sc[1], gap=sc[2], flags=flags)
                path = " ".join(c.last_path())
                assert ("f16=1" in path) == f16, (sc, flags, path)
                for k, (g, e) in enumerate(zip(got, exp)):
                    if flags:
                        assert (g["score"], g["end_x"], g["end_y"]) == (e["score"], e["end_x"], e["end_y"]), (sc, k, len(xs[k]), g, e)
                    else:
                        _cmp(g, e, "float16 pass, scoring %r, x[%d] |x|=%d" % (sc, k, len(xs[k])))
                beyond = c.last_counters()["beyond_f16"]
                if sc == (3.0, -3.0, 2.0):
                    # 129, 132, 180, 429 and the whole y (432); 123 and 126 are decided on float16 cells
                    assert beyond == 5, (flags, beyond)
                if not f16:
                    assert beyond == 0
        # a stream beyond the 16-bit step counters (pieces off): the float32 pass takes the launch
        c.set_option("no_wave_pieces", 1)
        long_x = [rnd(500, 70_000)[:30_000] + y[5:60] + rnd(501, 40_000), rnd(502, 200)]
        got = c.align_batch(long_x, y, semantics=0, flags=pgs.capi.SCORE_ONLY)
        assert "f16=0" in " ".join(c.last_path()) and "prof=1" in " ".join(c.last_path()), c.last_path()
        for g, x in zip(got, long_x):
            e = oracle.align(x, y, 0)
            assert (g["score"], g["end_x"], g["end_y"]) == (e["score"], e["end_x"], e["end_y"])
        c.set_option("no_wave_pieces", None)
    finally:
        c.close()
