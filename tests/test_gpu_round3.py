"""GPU parity, round 3: every A/B switch of the library lights an alternate kernel instance or pipeline that is also a
production fallback — each must give the oracle's answers on the fuzz case list; the pipelined long-query score kernel
(sw_long_kernel); the reference-sharding finish call (mi355_sw_align_scored_range)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pmap(fn, items, workers=8):
    """fn over items on a few threads (the oracle is a C library behind ctypes: its calls release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, items))


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
        singles.append((q, ref, sem, sc))
    singles = [a + (e,) for a, e in zip(singles, _pmap(lambda a: oracle.align(a[0], a[1], a[2], *a[3]), singles))]
    batches = []
    ref = pgs.synth.dna(5151, 70_000)
    refb = ref.tobytes()
    qs = [pgs.synth.read_from_ref(ref, 6000 + k, m, sub_rate=0.03, indel_rate=0.005)[0].tobytes()
          for k, m in enumerate((40, 100, 150, 150, 152, 300, 300, 600, 640, 1000, 1000, 2048, 2500, 2600, 700, 125))]
    qs += [b"", pgs.synth.dna(99, 150).tobytes()]
    for sem in (0, 1):
        for sc in ((3.0, -3.0, 2.0), (3.5, -3.25, 2.0)):
            batches.append((qs, refb, sem, sc, _pmap(lambda q: oracle.align(q, refb, sem, *sc), qs)))
    lone = []
    for k, m in enumerate((700, 1500, 2100, 3000)):
        q = pgs.synth.read_from_ref(ref, 6100 + k, m, sub_rate=0.02, indel_rate=0.004)[0].tobytes()
        for sem in (0, 1):
            lone.append((q, refb, sem, (3.0, -3.0, 2.0)))
    lone = [a + (e,) for a, e in zip(lone, _pmap(lambda a: oracle.align(a[0], a[1], a[2]), lone))]
    return singles, batches, lone


# every switch that selects another kernel instance / pipeline for the same answer
SWITCHES = ["no_f16", "no_unsat", "no_sample", "no_satflag", "no_solo", "no_wave", "no_comb", "no_twin", "no_wide",
            "no_strip", "no_quant", "no_devlist", "no_strip_groups", "u8_long_twin", "long_twin", "no_long",
            "no_requery", "slot=16", "strip_r=24", "few_r=5", "long_pipes=2", "no_long_p32", "long_groups=2", "force_f32", "no_opt_margin", "no_wave_prof",
            "no_wave_window", "no_first", "no_long_save", "assume_cus=32", "u8_sample_short", "no_wave_pieces", "no_u8_early", "no_wave_f16", "no_devlist_by_id"]


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
        if sem == 1:
            c.set_option("no_u8_early")                                          # (this test is about the SWEEP; the early exit: test_gpu_round4)
        variants = [{}, {"no_long_p32": 1}, {"no_long_p32": 1, "chunk": 8192}, {"long_pipes": 2}, {"long_groups": 2}, {"long_groups": 4, "long_pipes": 1},
                    {"long_wgs": 12}]
        cases = []                                                              # (query, expected): the oracle runs once per query
        for k, m in enumerate((2049, 2560, 2561, 3072, 5000, 7681, 10_000, 12_288)):
            o = [0, 150_000 - m, 70_000, 3000][k % 4]
            q = bytearray(refb[o:o + m])
            rng = np.random.default_rng(100 + k)
            for i in rng.choice(m, m // 50, replace=False):
                q[i] = b"ACGT"[int(rng.integers(0, 4))]
            cases.append(bytes(q).replace(b"N", b"A"))
        unrelated = pgs.synth.dna(7100 + sem, 4000).tobytes()                   # background maximum
        exps = _pmap(lambda q: oracle.align(q, refb, sem), cases + [unrelated])
        cases = list(zip(cases, exps[:-1]))
        unrelated_exp = exps[-1]
        for var in variants:
            for k, v in var.items():
                c.set_option(k, v)
            for q, exp in cases:
                m = len(q)
                got = c.align(q, refb, sem)
                _cmp(got, exp, "long kernel sem=%d m=%d %r" % (sem, m, var))
                name = c.last_kernel()["name"]
                if "no_long_p32" not in var or m <= 10_240:                     # (12 288 rows x 6 codes: float16 profile beyond one CU's LDS)
                    assert "sw_long_kernel" in name, name
                if "no_long_p32" in var and "sw_long_kernel" in name:
                    assert "f16 profile" in name, name
            _cmp(c.align(unrelated, refb, sem), unrelated_exp, "long kernel, unrelated query sem=%d %r" % (sem, var))
            for k in var:
                c.set_option(k, None)
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
                    exps = _pmap(lambda r: oracle.align(q, refb[r[0]:r[1]], sem), ranges)
                    for k, (lo, hi) in enumerate(ranges):
                        exp = exps[k]
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


def _drivers_dir():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "parallel-genomeseq_amd", "drivers")
    if not os.path.exists(os.path.join(d, "sw_solve_small.bin")):
        subprocess.check_call(["make", "-C", d])
    return d


def _eval_pos():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("eval_pos", os.path.join(root, "tools", "eval_pos.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_eval_report_on_data_small_matches_the_reference(data_small, tmp_path):
    """the reference's own evaluation (py/eval.py:102-121, `--option sw_solve_small`: rows whose
    pos_pred differs from the SAM POS) on the output of the sw_solve_small driver over the reference's data_small inputs:
    188 of 1170 positions differ with the float engine, 222 of 1170 with the uint8 engine (measured on the unmodified
    reference) — the greedy traceback overshoots, nothing is 'fixed' here."""
    import os
    import subprocess
    d = _drivers_dir()
    ev = _eval_pos()
    fa = tmp_path / "genome.fa"
    ref = data_small["ref"]
    fa.write_text(">22_5K\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    csv = tmp_path / "truth.csv"
    csv.write_text("index,QNAME,SEQ,POS\n" + "".join("%d,22_5K-1170,%s,%d\n" % (k, r, p) for k, (r, p) in
                                                       enumerate(zip(data_small["reads"], data_small["sam_pos"]))))
    for engine, want in (("f32", 188), ("u8", 222)):
        out = tmp_path / ("out_%s.csv" % engine)
        p = subprocess.run([os.path.join(d, "sw_solve_small.bin"), str(fa), str(csv), str(out), "--engine=" + engine],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode()
        n, bad = ev.count_mismatches(str(out))
        assert (n, bad) == (1170, want), (engine, n, bad)


def test_generator_driver_eval_chain(pgs, oracle, tmp_path):
    """tools/make_dataset.py (the reference's file shapes, py/ompfg_data_prep.py:92-116) -> drivers/sw_solve_small and
    drivers/sw_solve_big -> tools/eval_pos.py: every pos_pred / score of the output equals the oracle's, and the mismatch
    count of the report equals the one computed from the oracle's positions."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = _drivers_dir()
    ev = _eval_pos()
    data = tmp_path / "data"
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_dataset.py"), str(data), "--ref-len", "120000",
                           "--reads", "60", "--read-len", "150"])
    ref = (data / "custom_ref_1.fa").read_text().strip()
    rows = (data / "custom_reads_1.csv").read_text().splitlines()
    assert rows[0] == "index,QNAME,SEQ,POS" and len(rows) == 61
    reads = [r.split(",")[2] for r in rows[1:]]
    truth = [int(r.split(",")[3]) for r in rows[1:]]
    for engine, sem in (("f32", 0), ("u8", 1)):
        out = tmp_path / ("chain_%s.csv" % engine)
        p = subprocess.run([os.path.join(d, "sw_solve_small.bin"), str(data / "genome.fa"), str(data / "custom_reads_1.csv"), str(out),
                            "--engine=" + engine], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode()
        got = out.read_text().splitlines()[1:]
        exp = [oracle.align(q.encode(), ref.encode(), sem) for q in reads]
        for line, e in zip(got, exp):
            assert int(line.split(", ")[1]) == e["pos"] and float(line.split(", ")[2]) == e["score"], line[:40]
        n, bad = ev.count_mismatches(str(out))
        assert n == 60 and bad == sum(1 for e, t in zip(exp, truth) if e["pos"] != t), (engine, bad)
    # the benchmark driver on the same files: the reference's report lines, verbatim (sw_solve_big.cpp:71-74, :99-106)
    p = subprocess.run([os.path.join(d, "sw_solve_big.bin"), "2", "1", str(data / "custom_ref_1.fa"), str(data / "custom_reads_1.csv")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = p.stdout.decode()
    assert p.returncode == 0, text
    for want in ("[INFO] npiece: 2, nrepeat:1", "[INFO] Estimated Memory consumption 0.018GB", "[INFO] Theoretical GCUPS on Leonhard: ",
                 "[INFO] Average SW iter_ad_read times: ", ", GCPUS per iteration: ", "[INFO] GCUPS avg:", ", GCUPS std:"):
        assert want in text, (want, text)


@pytest.mark.parametrize("sem", [0, 1])
def test_per_query_fallback_on_repeat_rich_reference(pgs, oracle, sem):
    """Reads cut from poly-A runs, microsatellites and interspersed repeat families flag more candidate sub-chunks than their
    cap: THEY are swept again on the exact instances, the other reads of the batch keep the sampled sweep's candidates (no
    whole-batch second sweep).  Every result bit-exact against the oracle, with the per-query path, and with the whole-batch
    path it replaces (option no_requery)."""
    ref, planted = pgs.synth.dna_repeats(9001, 400_000, families=3, family_len=300, copies=40, tandem_runs=60, tandem_len=400,
                                         polya_runs=120, polya_len=300)
    reads, offs, which = pgs.synth.reads_with_repeats(ref, planted, 9002, 48, 150, repeat_fraction=0.2)
    assert len(which) >= 6
    refb = ref.tobytes()
    qs = [r.tobytes() for r in reads]
    exp = _pmap(lambda q: oracle.align(q, refb, sem), qs)
    c = pgs.Context(0)
    try:
        got = c.align_batch(qs, refb, semantics=sem)
        cnt = c.last_counters()
        for k, (g, e) in enumerate(zip(got, exp)):
            _cmp(g, e, "repeat-rich sem=%d read %d%s" % (sem, k, " (repeat)" if k in which else ""))
        # (reads that merely overlap a repeat copy may exceed their cap too — in the uint8 engine every hit of >= 83 bp saturates)
        assert cnt["whole_batch_again"] == 0 and cnt["requeried"] < len(qs) // 2, cnt
        if sem == 1:
            # uint8 engine: most offenders are settled by their first candidates in order (key at 255); without that step
            # (option no_first) every offender is swept again — same answers
            assert cnt["first_settled"] >= 1, cnt
            settled = cnt["first_settled"] + cnt["requeried"]
            c.set_option("no_first")
            got = c.align_batch(qs, refb, semantics=sem)
            cnt = c.last_counters()
            c.set_option("no_first", None)
            for k, (g, e) in enumerate(zip(got, exp)):
                _cmp(g, e, "repeat-rich, no_first, read %d" % k)
            assert cnt["first_settled"] == 0 and cnt["requeried"] == settled, (cnt, settled)
        else:
            assert 1 <= cnt["requeried"] and cnt["first_settled"] == 0, cnt
        c.set_option("no_requery")
        c.set_option("no_first")                                                # (else nobody is left over its cap in the uint8 engine)
        got = c.align_batch(qs, refb, semantics=sem)
        cnt = c.last_counters()
        for k, (g, e) in enumerate(zip(got, exp)):
            _cmp(g, e, "repeat-rich, whole batch again, sem=%d read %d" % (sem, k))
        assert cnt["whole_batch_again"] == 1 and cnt["requeried"] == 0, cnt
    finally:
        c.close()


def test_packed_upload_equals_pointer_upload(pgs, oracle):
    """mi355_sw_batch_upload_packed (one contiguous buffer + n + 1 offsets, the multi-FASTA shape of
    src/mpi_sw_solve_uniprot.cpp:97-110) == mi355_sw_batch_upload of the same sequences: UniProt-shaped batch against the
    144-aa query (every result vs the oracle), empty sequences, a buffer that does not start at offset 0, bad offsets."""
    n = 3000
    lens = pgs.synth.lognormal_lengths(15, n)
    lens[7] = 0
    lens[n - 1] = 0
    allres = pgs.synth.protein(15, int(lens.sum()) + 5)
    offs = np.concatenate([[5], 5 + np.cumsum(lens)]).astype(np.int64)      # the batch starts 5 bytes into the buffer
    seqs = [allres[offs[k]:offs[k + 1]].tobytes() for k in range(n)]
    query = pgs.synth.P02232
    c = pgs.Context(0)
    try:
        c.set_reference(query)
        c.batch_upload_packed(allres, offs)
        a = c.batch_run(semantics=0, raw=True)
        cons = [c.consensus(k) for k in range(0, n, 97)]
        c.batch_upload(seqs)
        b = c.batch_run(semantics=0, raw=True)
        for f in ("score", "pos", "end_x", "end_y", "cons_len"):
            assert np.array_equal(a[f], b[f]), f
        assert cons == [c.consensus(k) for k in range(0, n, 97)]
        for k in range(0, n, 131):
            e = oracle.align(seqs[k], query.encode(), 0)
            assert (a["score"][k], a["pos"][k], a["end_x"][k], a["end_y"][k]) == (e["score"], e["pos"], e["end_x"], e["end_y"]), k
        # reads against a long reference through the packed upload (score-kernel path)
        ref = pgs.synth.dna(16, 200_000)
        reads, _ = pgs.synth.fast_reads_from_ref(ref, 17, 33, 150)
        c.set_reference(ref)
        c.batch_upload_packed(reads.reshape(-1), np.arange(0, 34 * 150, 150, dtype=np.int64))
        got = c.batch_run(semantics=1)
        for r, g in zip(reads, got):
            _cmp(g, oracle.align(r.tobytes(), ref.tobytes(), 1), "packed reads")
        with pytest.raises(pgs.MI355Error):
            c.batch_upload_packed(allres, np.array([0, 10, 5], dtype=np.int64))
    finally:
        c.close()


def test_best_range_winner_only(pgs, oracle):
    """mi355_sw_best_range (what OMPParallelLocalAligner needs from the per-piece maxima, plocalaligner.cpp:122-129): winner
    and its maximum exact, first piece on ties, every piece that ties with the winner exact too; maxima of pieces that cannot
    win are lower bounds of unspecified slack, never above the truth (a lone long query is swept on the sampled maximum behind
    an optimistic warm-up margin: both only lower values); finishing the winner AND a loser through align_scored_range gives the piece's
    stand-alone alignment; short queries (exact sweep) behave like score_ranges."""
    ref = pgs.synth.dna(7401, 500_000)
    refb = bytearray(ref.tobytes())
    m = 3000
    q = bytes(refb[100_000:100_000 + m])
    refb[400_000:400_000 + m] = q                                   # the same hit twice: pieces 1 and 4 of 5 tie
    refb = bytes(refb)
    c = pgs.Context(0)
    try:
        c.set_reference(refb)
        c.batch_upload([q])
        for sem in (0, 1):
            ranges = pgs.capi.make_string_range(5, m, len(refb), 2.0)
            true = [oracle.score_only(q, refb[lo:hi], sem) for lo, hi in ranges]
            best, which, mx = c.best_range(ranges, semantics=sem)
            assert best[0] == max(true) and which[0] == true.index(max(true)), (sem, best, which, true)
            for k in range(5):
                assert mx[k, 0] <= true[k], (sem, k, mx[k, 0], true[k])
                if true[k] == max(true):
                    assert mx[k, 0] == true[k]
            if sem == 0:
                assert "sw_long_kernel" in c.last_kernel()["name"], c.last_kernel()["name"]
            for k in (int(which[0]), 2):
                lo, hi = ranges[k]
                _cmp(c.align_scored_range(k, semantics=sem), oracle.align(q, refb[lo:hi], sem), "best_range finish sem=%d piece %d" % (sem, k))
        # a short query: exact sweep, every maximum exact
        qs = bytes(refb[250_000:250_150])
        c.batch_upload([qs, q[:200]])
        ranges = pgs.capi.make_string_range(4, 200, len(refb), 2.0)
        best, which, mx = c.best_range(ranges, semantics=0)
        for j, qq in enumerate((qs, q[:200])):
            true = [oracle.score_only(qq, refb[lo:hi], 0) for lo, hi in ranges]
            assert mx[:, j].tolist() == true and best[j] == max(true) and which[j] == true.index(max(true))
    finally:
        c.close()


def test_optimistic_margin_is_certified_or_swept_again(pgs, oracle):
    """sw_long_kernel's optimistic warm-up margin (exact for maxima above 11/12 of the best possible score): a query with a
    strong hit is certified by one sweep; a diverged copy (25 % substitutions: its score lies below what the margin certifies)
    and a hit that crosses tile borders make the library sweep again with the margin that score needs — results bit-exact
    either way, for the whole-reference call, for best_range on its own, and through the multi-rank protocol (exact_above /
    known_best) emulated with two contexts holding the even and the odd pieces."""
    ref = pgs.synth.dna(7501, 400_000)
    m = 4000
    rng = np.random.default_rng(7502)
    strong = bytearray(ref[150_000:150_000 + m].tobytes())
    for i in rng.choice(m, m // 100, replace=False):
        strong[i] = b"ACGT"[int(rng.integers(0, 4))]
    weak = bytearray(ref[300_000:300_000 + m].tobytes())
    for i in rng.choice(m, m // 4, replace=False):
        weak[i] = b"ACGT"[int(rng.integers(0, 4))]
    refb = ref.tobytes()
    c = pgs.Context(0)
    c2 = pgs.Context(0)
    try:
        for name, q in (("strong", bytes(strong)), ("weak", bytes(weak))):
            exp = oracle.align(q, refb, 0)
            got = c.align(q, refb, 0)
            _cmp(got, exp, "optimistic margin, whole reference, %s hit" % name)
            again = c.last_counters()["whole_batch_again"]
            assert again == (0 if name == "strong" else 1), (name, again, exp["score"])
            # reference sharding: ranges and the oracle's per-piece answers
            ranges = pgs.capi.make_string_range(6, m, len(refb), 2.0)
            true = _pmap(lambda r: oracle.score_only(q, refb[r[0]:r[1]], 0), ranges)
            c.set_reference(refb); c.batch_upload([q])
            best, which, mx = c.best_range(ranges, semantics=0)
            assert best[0] == max(true) and which[0] == true.index(max(true)), (name, best, which, true)
            _cmp(c.align_scored_range(int(which[0]), semantics=0), oracle.align(q, refb[ranges[which[0]][0]:ranges[which[0]][1]], 0), "finish %s" % name)
            # two "ranks": even / odd pieces, merged as dist.align_split_sharded_certified does
            c2.set_reference(refb); c2.batch_upload([q])
            shares = [(c, list(range(0, 6, 2))), (c2, list(range(1, 6, 2)))]
            known, rounds = 0.0, 0
            for rounds in (1, 2, 3):
                keys, above = [], -1.0
                for cc, pieces in shares:
                    b, w, _, a = cc.best_range([ranges[p] for p in pieces], semantics=0, known_best=known, want_exact_above=True)
                    keys.append((float(b[0]), -pieces[int(w[0])]))
                    above = max(above, a)
                gbest, negpiece = max(keys)
                if gbest > above:
                    break
                known = max(gbest, 1.0)
            assert gbest == max(true) and -negpiece == true.index(max(true)), (name, gbest, negpiece, true)
            assert rounds == (1 if name == "strong" else 2), (name, rounds)
        # the option that switches the optimistic margin off: one sweep with the full margin
        c.set_option("no_opt_margin")
        _cmp(c.align(bytes(weak), refb, 0), oracle.align(bytes(weak), refb, 0), "full margin")
        assert c.last_counters()["whole_batch_again"] == 0
    finally:
        c.close()
        c2.close()


@pytest.mark.parametrize("ylen", [144, 300, 500])
def test_small_alignment_batch_windows(pgs, oracle, ylen):
    """The many-small-alignments batch with traceback on sw_wave_prof_kernel: checkpointed first pass + decisions for a window in
    front of each argmax (host_batch.h).  Database sequences that make the walk do everything it can: unrelated ones (short walks),
    the second sequence planted at the start / in the middle / at the very end of a long x (argmax inside the first window,
    across a checkpoint, walks LONGER than a window -> the host-driven path takes the problem), mutated copies with indels,
    lengths around the 64-row checkpoints, empty and one-letter sequences.  Every field against the oracle, and the same call
    with whole-problem decisions (option no_wave_window) and with the compare-and-select kernel (no_wave_prof)."""
    rng = np.random.default_rng(4242 + ylen)
    y = pgs.synth.protein(77 + ylen, ylen).tobytes()
    xs = [b"", b"A", y[:1], y, y[:40], y[-40:]]
    for m in (1, 2, 15, 16, 17, 63, 64, 65, 111, 112, 113, 127, 128, 129, 191, 192, 193, 255, 256, 257, 400, 1000, 3000):
        xs.append(pgs.synth.protein(9000 + m, m).tobytes())                     # unrelated
    for k, m in enumerate((300, 700, 1500, 4000)):
        base = bytearray(pgs.synth.protein(9500 + k, m).tobytes())
        cut = [y, y[: ylen // 2], y[ylen // 3:], y[10:60]][k % 4]
        for at in (0, m // 2 - 7, m - len(cut)):
            x = bytearray(base)
            x[at:at + len(cut)] = cut
            xs.append(bytes(x))
            # a diverged copy with substitutions and a few indels: a long walk with gaps
            c = bytearray(cut)
            for i in rng.choice(len(c), max(1, len(c) // 12), replace=False):
                c[i] = b"ACDEFGHIKLMNPQRSTVWY"[int(rng.integers(0, 20))]
            for i in sorted(rng.choice(len(c) - 2, 3, replace=False), reverse=True):
                if rng.random() < 0.5:
                    del c[i]
                else:
                    c.insert(i, b"ACDEFGHIKLMNPQRSTVWY"[int(rng.integers(0, 20))])
            x = bytearray(base)
            x[at:at + len(c)] = c
            xs.append(bytes(x[:m]) if at + len(c) > m else bytes(x))
    xs.append((y * 6)[:5 * ylen + 17])                                          # tandem copies: ties between equal maxima
    exp = [oracle.align(x, y, 0) for x in xs]
    c = pgs.Context(0)
    try:
        for var in ({}, {"no_wave_window": 1}, {"no_wave_prof": 1}, {"no_wave_f16": 1}):
            for k, v in var.items():
                c.set_option(k, v)
            got = c.align_batch(xs, y, semantics=0)
            for k, (g, e) in enumerate(zip(got, exp)):
                _cmp(g, e, "windows |y|=%d x[%d] |x|=%d %r" % (ylen, k, len(xs[k]), var))
            cnt = c.last_counters()
            left = cnt["left_window"]
            # the planted copies walk further than a window — where they get one (beyond the float16 pass's key range they do not)
            # (on the float16 pass a diverged copy below 128 may still walk further than its window: not counted on either way)
            windows_f32 = var == {"no_wave_f16": 1} or (not var and ylen != 144)
            windows_f16 = not var and ylen == 144
            assert windows_f16 or ((left >= 8) if windows_f32 else (left == 0)), (var, left)
            # ten columns per lane (|y| = 144): the first pass runs on packed float16 cells, and the planted copies (scores of
            # 128 and more) are beyond its key range: they must come back through the float32 path with the oracle's answers
            f16 = "f16=1" in " ".join(c.last_path())
            assert f16 == (ylen == 144 and not var), (var, c.last_path())
            assert (cnt["beyond_f16"] >= 8) if f16 else (cnt["beyond_f16"] == 0), (var, cnt)
            for k in var:
                c.set_option(k, None)
        # score + first maximum only (no decisions anywhere: the float16 pass on every |y| = 144 sequence, pieces included)
        got = c.align_batch(xs, y, semantics=0, flags=pgs.capi.SCORE_ONLY)
        assert ("f16=1" in " ".join(c.last_path())) == (ylen == 144), c.last_path()
        for k, (g, e) in enumerate(zip(got, exp)):
            assert (g["score"], g["end_x"], g["end_y"]) == (e["score"], e["end_x"], e["end_y"]), ("score only", ylen, k, len(xs[k]), g, e)
        # other dyadic scorings on the same batch (other scale; a cheap gap: long gapped walks)
        for sc in ((2.0, -1.0, 0.5), (5.0, -4.0, 3.0)):
            for g, x in zip(c.align_batch(xs, y, semantics=0, match=sc[0], mismatch=sc[1], gap=sc[2]), xs):
                _cmp(g, oracle.align(x, y, 0, *sc), "windows |y|=%d |x|=%d scoring %r" % (ylen, len(x), sc))
    finally:
        c.close()


def test_first_candidates_and_the_wrapped_triangle(pgs, oracle):
    """uint8 engine, reads over their candidate cap settled by their first candidates in order (sw_sample_first): the skewed
    storage order wraps the bottom-right triangle (i + j > |y|) IN FRONT of every other cell, so a 255 in the reference's last
    |x| columns beats the first copy — the last two sub-chunks are always evaluated.  A 150 bp element planted 160 times, one copy
    ending exactly at the end of the reference, one at its start; reads = the element, diverged copies, unrelated reads.
    (Found by tests/stress.py: seed 2026.)"""
    n, m = 400_000, 150
    ref = bytearray(pgs.synth.dna(8801, n).tobytes())
    elem = pgs.synth.dna(8802, m).tobytes()
    for k in range(158):
        at = 3000 + 2500 * k
        ref[at:at + m] = elem
    ref[n - m:] = elem                                                          # ends with the reference: the wrapped triangle
    ref[0:m] = elem
    refb = bytes(ref)
    rng = np.random.default_rng(8803)
    reads = [elem, elem[10:] + b"ACGTACGTAC", elem[:120] + pgs.synth.dna(8804, 30).tobytes()]
    for k in range(5):
        q = bytearray(elem)
        for i in rng.choice(m, 6, replace=False):
            q[i] = b"ACGT"[int(rng.integers(0, 4))]
        reads.append(bytes(q))
    reads += [pgs.synth.dna(8810 + k, m).tobytes() for k in range(24)]
    exp = _pmap(lambda q: oracle.align(q, refb, 1), reads)
    assert exp[0]["end_y"] > n - m, exp[0]                                      # the oracle's first 255 IS in the wrapped triangle
    c = pgs.Context(0)
    try:
        got = c.align_batch(reads, refb, semantics=1)
        cnt = c.last_counters()
        for k, (g, e) in enumerate(zip(got, exp)):
            _cmp(g, e, "wrapped triangle, read %d" % k)
        assert cnt["first_settled"] >= 4, cnt
        # the same reference without the copy at the end: the first copy (at the reference's start) wins
        ref2 = refb[:n - m] + pgs.synth.dna(8805, m).tobytes()
        for k, (g, q) in enumerate(zip(c.align_batch(reads[:8], ref2, semantics=1), reads[:8])):
            _cmp(g, oracle.align(q, ref2, 1), "no copy at the end, read %d" % k)
    finally:
        c.close()


def test_lone_long_query_inside_a_batch(pgs, oracle):
    """A batch whose only long query (2300 rows) has a bucket of its own runs it on sw_long_kernel — with the PROVEN warm-up margin:
    the optimistic margin is certified for single-query calls only.  The case is the one tests/stress.py (seed 777) found: scoring
    7 / -7 / 1 (cheap gaps: the best alignment spans 5963 columns, far beyond the optimistic margin), 20 000 columns, sixteen
    queries of 0 .. 2300 bp (tests/golden/stress_seed777_long_query_in_batch.npz holds its bytes)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stress_seed777_long_query_in_batch.npz"))
    ref = d["ref"].tobytes()
    offs = np.concatenate([[0], np.cumsum(d["lens"])])
    allq = d["qs"].tobytes()
    qs = [allq[offs[k]:offs[k + 1]] for k in range(len(d["lens"]))]
    sc = tuple(float(v) for v in d["sc"])
    exp = _pmap(lambda q: oracle.align(q, ref, 0, *sc), qs)
    c = pgs.Context(0)
    try:
        for var in ({}, {"no_sample": 1}, {"long_pipes": 1}, {"chunk": 4096}):
            for k, v in var.items():
                c.set_option(k, v)
            for k, (g, e) in enumerate(zip(c.align_batch(qs, ref, semantics=0, match=sc[0], mismatch=sc[1], gap=sc[2]), exp)):
                _cmp(g, e, "long query in a batch, query %d (%d bp) %r" % (k, len(qs[k]), var))
            assert "sw_long_kernel" in c.last_kernel()["name"], c.last_kernel()["name"]
            for k in var:
                c.set_option(k, None)
        # the same query alone: optimistic margin, not certified (score 10362 of 16100), swept again
        _cmp(c.align(qs[6], ref, 0, *sc), exp[6], "the long query alone")
        assert c.last_counters()["whole_batch_again"] == 1
    finally:
        c.close()


@pytest.mark.parametrize("sc", [(1.1, -0.9, 0.37), (3.0, -3.0, 0.5)])
def test_uint8_gap_truncated_to_zero_and_too_long_for_lds(pgs, oracle, sc):
    """uint8 engine, a gap penalty that truncates to 0 (_saturate, similaritymatrix.cpp:376-384): no finite warm-up margin, so the
    problem never takes the score kernel; with an anti-diagonal that does not fit the LDS kernel either it used to return ENOTSUP
    (tests/stress.py seed 777: 17 000 x 17 647 at 1.1 / -0.9 / 0.37).  It runs on the strip kernel over the whole range as ONE
    window (a window that starts at column 0 needs no margin): every field against the oracle, alone and inside a batch."""
    m, n = 14_000, 14_500
    ref = pgs.synth.protein(9901, n).tobytes()
    q = bytearray(ref[200:200 + m])
    rng = np.random.default_rng(9902)
    for i in rng.choice(m, m // 10, replace=False):
        q[i] = b"ACDEFGHIKLMNPQRSTVWY"[int(rng.integers(0, 20))]
    del q[5000:5040]
    q = bytes(q)
    others = [q[:300], q[:4000], b"", pgs.synth.protein(9903, 700).tobytes()]
    exps = _pmap(lambda x: oracle.align(x, ref, 1, *sc), [q] + others)
    c = pgs.Context(0)
    try:
        _cmp(c.align(q, ref, 1, *sc), exps[0], "gap truncated to 0 %r" % (sc,))
        got = c.align_batch([q] + others, ref, semantics=1, match=sc[0], mismatch=sc[1], gap=sc[2])
        for g, e, x in zip(got, exps, [q] + others):
            _cmp(g, e, "gap truncated to 0 %r, in a batch, |x| = %d" % (sc, len(x)))
    finally:
        c.close()
