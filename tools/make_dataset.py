#!/usr/bin/env python3
"""Synthetic data in the file shapes the reference's drivers consume (reference
py/ompfg_data_prep.py:92-116 writes the same shapes from chr22):

  <out>/custom_ref_1.fa      single headerless line            (src/sw_solve_big.cpp:30-37)
  <out>/custom_reads_1.csv   header index,QNAME,SEQ,POS + rows  (src/sw_solve_big.cpp:69, sw_solve_small.cpp:56-67)
  <out>/genome.fa            same reference as FASTA with one header line, 60 columns (sw_solve_small.cpp:20-31)

POS is the 1-based position the read was cut from (the SAM-style ground truth of data_small_ground_truth.csv).
Deterministic: SplitMix64 seeds as in SURVEY.md §8d."""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_synth():
    spec = importlib.util.spec_from_file_location("pgs_synth", os.path.join(ROOT, "parallel-genomeseq_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--ref-len", type=int, default=1_000_000)
    ap.add_argument("--reads", type=int, default=100)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--ref-seed", type=int, default=1)
    ap.add_argument("--read-seed", type=int, default=2)
    a = ap.parse_args(argv)
    synth = load_synth()
    os.makedirs(a.out, exist_ok=True)
    ref = synth.dna(a.ref_seed, a.ref_len)
    refs = ref.tobytes().decode()
    with open(os.path.join(a.out, "custom_ref_1.fa"), "w") as f:
        f.write(refs + "\n")
    with open(os.path.join(a.out, "genome.fa"), "w") as f:
        f.write(">synthetic_%d\n" % a.ref_len)
        for i in range(0, len(refs), 60):
            f.write(refs[i:i + 60] + "\n")
    reads, offs = synth.reads_from_ref(ref, a.read_seed, a.reads, a.read_len)
    with open(os.path.join(a.out, "custom_reads_1.csv"), "w") as f:
        f.write("index,QNAME,SEQ,POS\n")
        for k in range(a.reads):
            f.write("%d,synth-%d,%s,%d\n" % (k, k, reads[k].tobytes().decode(), int(offs[k]) + 1))
    print("wrote %s: reference %d bp, %d reads x %d bp" % (a.out, a.ref_len, a.reads, a.read_len))


if __name__ == "__main__":
    main()
