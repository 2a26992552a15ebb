"""Handle on oracle/_ref/ref_driver — the REAL reference aligner behind our command driver.

TEST INFRASTRUCTURE ONLY.  Exists in this container (built by oracle/build_ref.sh from
/root/reference); the prebuilt binaries also travel to the GPU box, where only bench.py's
cpu_baseline leg uses ref_driver_omp.  Tests that need it skip when it is absent.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(_HERE, "_ref", "ref_driver")
DRIVER_OMP = os.path.join(_HERE, "_ref", "ref_driver_omp")
SEM = {0: "f32", 1: "u8"}


def available():
    return os.access(DRIVER, os.X_OK)


def run(commands, omp=False, env=None, timeout=600):
    """Send command lines; return reply lines (one per command)."""
    exe = DRIVER_OMP if omp else DRIVER
    p = subprocess.run([exe], input=("\n".join(commands) + "\n").encode("latin-1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("ref_driver failed rc=%d: %s" % (p.returncode, p.stderr.decode()[-400:]))
    return p.stdout.decode("latin-1").splitlines()


def parse_align(line):
    t = line.split(" ")
    cx = "" if t[4] == "*" else t[4]
    cy = "" if t[5] == "*" else t[5]
    return dict(score=float(t[0]), pos=int(t[1]), end_x=int(t[2]), end_y=int(t[3]), cons_x=cx, cons_y=cy)


def align_cmd(x, y, semantics=0, match=3.0, mismatch=-3.0, gap=2.0):
    return "align %s %r %r %r %s %s" % (SEM[semantics], match, mismatch, gap, x, y)


def alignlut_cmd(x, y, semantics, seed, scale, gap):
    return "alignlut %s %d %r %r %s %s" % (SEM[semantics], seed, scale, gap, x, y)


def split_cmd(x, y, sm_sem, la_sem, npiece, ratio, match=3.0, mismatch=-3.0, gap=2.0):
    return "split %s %s %r %r %r %d %r %s %s" % (SEM[sm_sem], SEM[la_sem], match, mismatch, gap, npiece, ratio, x, y)
