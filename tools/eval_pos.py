#!/usr/bin/env python3
"""Position-mismatch report of the reference's py/eval.py:102-121 (`--option sw_solve_small`): counts rows of a
driver output CSV (`<input_line>, <pos_pred>, <score>`) whose pos_pred differs from the POS column.
Mismatches are expected behaviour of the reference's greedy traceback, nothing asserts."""
import sys


def count_mismatches(path):
    n = bad = 0
    with open(path) as f:
        header = f.readline().rstrip("\n").split(",")
        ipos = header.index("POS")
        ipred = header.index("pos_pred")
        for line in f:
            t = [x.strip() for x in line.rstrip("\n").split(",")]
            n += 1
            bad += int(t[ipos]) != int(t[ipred])
    return n, bad


if __name__ == "__main__":
    n, bad = count_mismatches(sys.argv[1] if len(sys.argv) > 1 else "data/align_output.csv")
    print("%d of %d positions differ from POS. May be caused by cost function" % (bad, n))
