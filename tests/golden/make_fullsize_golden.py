#!/usr/bin/env python3
"""make_fullsize_golden.py — FULL-SIZE fixtures from the REAL reference (oracle/_ref/ref_driver = the unmodified
src/aligner/*.cpp of /root/reference behind oracle/ref_driver.cpp).  Runs in the build container only; the GPU tests
compare against the committed tests/golden/fullsize.json (SURVEY.md §8(d) cfg 3 / cfg 4).

  * configs[2] shape, 150 bp x 50 Mbp: the reference itself holds one alignment at a time (Similarity_Matrix: 30 GB of
    float cells, Similarity_Matrix_Skewed: 9.2 GB) — 16 float-engine and 64 uint8-engine alignments, 8 / 4 of them reads
    cut from inside the planted repeats of synth.dna_repeats (seed 33; the bench's repeat_rich reference), the others
    synth.reads_from_ref (1 % substitutions, 0.1 % indels, seed 4) of synth.dna(3, 50 Mbp);
  * configs[3] shape: all 561 356 UniProt-shaped sequences (synth.lognormal_lengths(5) / synth.protein(5)) as FIRST argument
    against the P02232 query, SWAligner<Similarity_Matrix>, default scoring (src/mpi_sw_solve_uniprot.cpp:120): sha256 over
    the lines "score pos\n" in database order, plus the first 32 and every 10 007th result in the clear.

usage: make_fullsize_golden.py [--skip-f32] [--skip-u8] [--skip-c4] [--procs N]   (≈ 25 min, ≤ 31 GB of memory)
"""
import argparse
import hashlib
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "fullsize.json")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def load_package():
    name = "parallel_genomeseq_amd"
    path = os.path.join(ROOT, "parallel-genomeseq_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def drive(commands, timeout):
    p = subprocess.run([DRIVER], input=("\n".join(commands) + "\n").encode("latin-1"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("ref_driver rc=%d: %s" % (p.returncode, p.stderr.decode()[-400:]))
    return p.stdout.decode("latin-1").splitlines()


def parse(line):
    t = line.split(" ")
    return dict(score=float(t[0]), pos=int(t[1]), end_x=int(t[2]), end_y=int(t[3]),
                cons_x="" if t[4] == "*" else t[4], cons_y="" if t[5] == "*" else t[5])


# what the fixture and tests/test_gpu_fullsize.py agree on (the test regenerates the same inputs from these numbers)
REF_LEN = 50_000_000
PLAIN = dict(seed=3)
REPEATS = dict(seed=33, families=4, family_len=300, copies=4000, divergence=0.03, tandem_runs=500, tandem_len=400,
               polya_runs=500, polya_len=300)
READ_SEED, READ_LEN = 4, 150
N_F32_PLAIN, N_F32_REP, N_U8_PLAIN, N_U8_REP = 12, 4, 56, 8
REP_READ_SEED = 34


def config3_reads(pgs):
    """(plain reference, its reads), (repeat reference, reads cut from inside its repeats)."""
    ref = pgs.synth.dna(PLAIN["seed"], REF_LEN)
    reads, _ = pgs.synth.reads_from_ref(ref, READ_SEED, N_U8_PLAIN, READ_LEN)
    rep, planted = pgs.synth.dna_repeats(REPEATS["seed"], REF_LEN, **{k: v for k, v in REPEATS.items() if k != "seed"})
    rr, _, which = pgs.synth.reads_with_repeats(rep, planted, REP_READ_SEED, 2048, READ_LEN, repeat_fraction=0.01)
    rep_reads = [rr[i] for i in which[:N_U8_REP]]
    return (ref, [r.tobytes().decode() for r in reads]), (rep, [r.tobytes().decode() for r in rep_reads])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-f32", action="store_true")
    ap.add_argument("--skip-u8", action="store_true")
    ap.add_argument("--skip-c4", action="store_true")
    ap.add_argument("--procs", type=int, default=6, help="processes of the configs[3] pass (a few MB each)")
    args = ap.parse_args()
    if not os.access(DRIVER, os.X_OK):
        raise SystemExit("oracle/_ref/ref_driver missing: run oracle/build_ref.sh (needs /root/reference)")
    pgs = load_package()
    fix = json.load(open(OUT)) if os.path.exists(OUT) else {}
    fix["generator"] = "tests/golden/make_fullsize_golden.py against oracle/_ref/ref_driver (the unmodified reference)"
    fix["config3"] = fix.get("config3", {})
    c3 = fix["config3"]
    c3.update(ref_len=REF_LEN, plain=PLAIN, repeats=REPEATS, read_seed=READ_SEED, read_len=READ_LEN,
              repeat_read_seed=REP_READ_SEED, scoring=[3.0, -3.0, 2.0])
    tmp = tempfile.mkdtemp(prefix="fullsize_", dir="/tmp")
    t_all = time.time()
    if not (args.skip_f32 and args.skip_u8):
        (ref, reads), (rep, rep_reads) = config3_reads(pgs)
        c3["plain_reads"] = reads
        c3["repeat_reads"] = rep_reads
        for name, arr in (("plain", ref), ("repeats", rep)):
            with open(os.path.join(tmp, name + ".txt"), "wb") as f:
                f.write(arr.tobytes() + b"\n")
        del ref, rep
        for sem, skip, npl, nrp in (("u8", args.skip_u8, N_U8_PLAIN, N_U8_REP), ("f32", args.skip_f32, N_F32_PLAIN, N_F32_REP)):
            if skip:
                continue
            for name, rs, cnt in (("plain", reads, npl), ("repeats", rep_reads, nrp)):
                t0 = time.time()
                cmds = ["loadref " + os.path.join(tmp, name + ".txt")] + ["alignref %s 3.0 -3.0 2.0 %s" % (sem, r) for r in rs[:cnt]]
                out = drive(cmds, timeout=3 * 3600)
                assert out[0] == "loaded %d" % REF_LEN, out[0]
                c3["%s_%s" % (sem, name)] = [parse(l) for l in out[1:]]
                assert len(c3["%s_%s" % (sem, name)]) == cnt
                print("config3 %s %s: %d alignments in %.0f s" % (sem, name, cnt, time.time() - t0), flush=True)
                json.dump(fix, open(OUT, "w"), indent=0)
    if not args.skip_c4:
        nseq = 561_356
        lens = pgs.synth.lognormal_lengths(5, nseq)
        res = pgs.synth.protein(5, int(lens.sum()))
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        t0 = time.time()
        bounds = [nseq * k // args.procs for k in range(args.procs + 1)]
        procs = []
        for k in range(args.procs):
            path = os.path.join(tmp, "db%d.txt" % k)
            with open(path, "wb") as f:
                for s in range(bounds[k], bounds[k + 1]):
                    f.write(res[offs[s]:offs[s + 1]].tobytes() + b"\n")
            procs.append(subprocess.Popen([DRIVER], stdin=subprocess.PIPE, stdout=subprocess.PIPE))
            procs[-1].stdin.write(("manyfirst %s %s\n" % (path, pgs.synth.P02232)).encode())
            procs[-1].stdin.close()
        lines = []
        for k, p in enumerate(procs):
            out = p.stdout.read().decode().splitlines()
            assert p.wait() == 0 and out[-1] == "done %d" % (bounds[k + 1] - bounds[k]), (k, out[-1:])
            lines += out[:-1]
        assert len(lines) == nseq
        h = hashlib.sha256()
        for l in lines:
            sc, pos = l.split(" ")
            h.update(("%.9g %d\n" % (float(sc), int(pos))).encode())
        fix["config4"] = dict(sequences=nseq, seed=5, query="P02232", scoring=[3.0, -3.0, 2.0], semantics="f32",
                              line_format="%.9g %d\\n (score, pos) in database order", sha256=h.hexdigest(),
                              first=[[float(l.split()[0]), int(l.split()[1])] for l in lines[:32]],
                              every_10007th=[[float(lines[i].split()[0]), int(lines[i].split()[1])] for i in range(0, nseq, 10007)],
                              sum_score=float(sum(float(l.split()[0]) for l in lines)), sum_pos=int(sum(int(l.split()[1]) for l in lines)))
        print("config4: %d alignments in %.0f s, sha256 %s" % (nseq, time.time() - t0, h.hexdigest()), flush=True)
    json.dump(fix, open(OUT, "w"), indent=0)
    for f in os.listdir(tmp):
        os.unlink(os.path.join(tmp, f))
    os.rmdir(tmp)
    print("wrote %s in %.0f s" % (OUT, time.time() - t_all))


if __name__ == "__main__":
    main()
