"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/mi355_sw.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mi355_sw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_sw_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(pgs):
    L = pgs.capi.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "symbol %s declared in include/mi355_sw.h is not exported" % n
    assert sorted(pgs.capi.EXPORTS) == names
    assert b"gfx950" in L.mi355_sw_build_info()


def test_make_string_range_host_helper(pgs, golden):
    for c in golden["range"]:
        assert pgs.capi.make_string_range(c["npiece"], c["short"], c["long"], c["ratio"]) == [tuple(r) for r in c["ranges"]]
    assert pgs.capi.make_string_range(4, 100, 120, 2.0) is None     # assert overlap <= piecelen would fire


def test_no_cpu_fallback(pgs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pgs.MI355Error):
        pgs.Context(0)
    with pytest.raises(pgs.MI355Error):
        pgs.SWAligner("GGTTGACTA", "TGTTACGG").calculateScore()


def test_aligner_mirror_defaults(pgs):
    la = pgs.SWAligner("GGTTGACTA", "TGTTACGG")
    assert la.getScore() == -1.0 and la.getPos() == 0 and la.getConsensus_x() == ""   # smithwaterman.cpp:27-33
    with pytest.raises(AssertionError):
        pgs.OMPParallelLocalAligner("A" * 100, "C" * 120, 4, 2.0)                     # plocalaligner.cpp:52


def test_cpp_dropin_headers_compile():
    """include/parseq/*.h (the C++ mirror of the reference classes) compile and link against the C-ABI — with the
    POD fallbacks, and (where the reference's vendored Eigen zip is available) with the reference's Eigen signatures:
    Eigen::VectorXf getTimings(), const Eigen::MatrixXf &get_matrix(), const MatrixX8u &get_matrix()."""
    import subprocess
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "cpp", "build_dropin.sh")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_dropin.bin"))
    if os.path.exists("/root/reference/cmake/eigen-3.3.7.zip"):
        assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_dropin_eigen.bin"))
        # the reference's own test/*.cpp, unchanged and in place, against the mirror + tests/cpp/gtest_shim
        assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "test_reference_gtests.bin"))


def test_index_maps_match_reference(pgs, golden):
    """mi355_sw_true2raw / raw2true against the real reference's trueindex2rawindex (tests/golden) and
    the round trip of test/test_skewedmatrix.cpp:5-37."""
    for c in golden["true2raw"]:
        m, n, k = c["m"], c["n"], 0
        for ti in range(n + 1):
            for tj in range(m + 1):
                raw = pgs.capi.true2raw(m, n, ti, tj)
                assert list(raw) == c["raw"][k]
                assert pgs.capi.raw2true(m, n, *raw) == (ti, tj)
                k += 1


def test_dataset_tools(tmp_path):
    """tools/make_dataset.py writes the reference drivers' file shapes; tools/eval_pos.py reproduces the
    mismatch count of py/eval.py:112-121 on a driver-shaped CSV."""
    import importlib.util
    for name in ("make_dataset", "eval_pos"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        globals()[name] = mod
    make_dataset.main([str(tmp_path), "--ref-len", "5000", "--reads", "7", "--read-len", "50"])
    ref = open(tmp_path / "custom_ref_1.fa").read().split("\n")
    assert len(ref[0]) == 5000 and set(ref[0]) <= set("ACGT")
    rows = open(tmp_path / "custom_reads_1.csv").read().splitlines()
    assert rows[0] == "index,QNAME,SEQ,POS" and len(rows) == 8 and len(rows[1].split(",")[2]) == 50
    fa = open(tmp_path / "genome.fa").read().splitlines()
    assert fa[0].startswith(">") and "".join(fa[1:]) == ref[0]
    out = tmp_path / "out.csv"
    out.write_text("index,QNAME,SEQ,POS,pos_pred,score\n0,a,ACGT,10, 10, 12\n1,b,ACGT,20, 17, 12\n")
    assert eval_pos.count_mismatches(str(out)) == (2, 1)


def test_worker_pool_under_thread_sanitizer(tmp_path):
    """csrc/host_common.h's WorkerPool (spin-then-sleep workers behind parallel_for: the per-item host loops of a big batch) built
    on its own with g++ -fsanitize=thread and stressed by tests/cpp/worker_pool_stress.cpp: every part of every run exactly once,
    no data race reported."""
    import subprocess
    src = open(os.path.join(ROOT, "parallel-genomeseq_amd", "csrc", "host_common.h")).read()
    a = src.index("class WorkerPool {")
    b = src.index("template <class F>\nvoid parallel_for(size_t n, F fn")
    head = "\n".join("#include <%s>" % h for h in ("algorithm", "atomic", "chrono", "condition_variable", "cstdint", "functional", "mutex", "thread", "vector"))
    cpp = tmp_path / "pool.cpp"
    cpp.write_text(head + "\n" + src[a:b] + open(os.path.join(ROOT, "tests", "cpp", "worker_pool_stress.cpp")).read())
    exe = tmp_path / "pool"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", str(cpp), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok") and "ThreadSanitizer" not in r.stderr, (r.stdout[-500:], r.stderr[-2000:])
