"""The randomised differential runs (tests/stress.py, tests/stress_small_batches.py) as collected GPU tests: three
committed seeds each, time-boxed, every case bit-compared with the CPU oracle (score, pos, argmax cell, both consensus
strings).  Two of round 3's three findings came from exactly these loops; run by hand their logs never reached the
driver.  A failure prints the mismatching cases (a ragged batch is saved under gpurun_out/ for tests/replay_case.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

BUDGET_S = float(os.environ.get("MI355_SW_STRESS_SECONDS", "32"))


@pytest.mark.parametrize("seed", [2026, 777, 424242])
def test_stress_mixed_shapes(seed):
    """Single alignments up to 17 k rows (tables, fractional and non-dyadic scorings, seven alphabets), ragged batches, long
    float-engine batches on the saturating sweep, many small whole problems, the split aligner, both engines."""
    import stress
    ncase, nbad = stress.run(BUDGET_S, seed)
    assert ncase > 0 and nbad == 0, "seed %d: %d mismatches in %d cases\n%s" % (seed, nbad, ncase, "\n".join(stress.findings[:20]))


@pytest.mark.parametrize("seed", [4321, 99, 31337])
def test_stress_small_alignment_batches(seed):
    """The many-small-alignments batch (UniProt shape: shared-profile kernel, checkpointed decision windows, walks that leave
    their window, non-dyadic scorings on sw_wave_kernel): |y| 1-512, six alphabets, fourteen scorings."""
    import stress_small_batches as ssb
    ncase, nbad, nleft = ssb.run(BUDGET_S, seed)
    assert ncase > 0 and nbad == 0, "seed %d: %d mismatches in %d alignments\n%s" % (seed, nbad, ncase, "\n".join(ssb.findings[:20]))
