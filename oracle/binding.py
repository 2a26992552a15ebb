"""ctypes binding of oracle/libsw_oracle.so (the C restatement, sw_oracle.c).

TEST INFRASTRUCTURE ONLY — the checker, never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F32, U8SAT = 0, 1


class Scoring(C.Structure):
    _fields_ = [("lut", C.POINTER(C.c_float)), ("match", C.c_float), ("mismatch", C.c_float),
                ("gap", C.c_float)]


class Result(C.Structure):
    _fields_ = [("score", C.c_float), ("pos", C.c_uint32), ("end_x", C.c_int64), ("end_y", C.c_int64),
                ("cons_x", C.c_void_p), ("cons_y", C.c_void_p), ("cons_len", C.c_size_t)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsw_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsw_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.sw_oracle_align.restype = C.c_int
        L.sw_oracle_align_split.restype = C.c_int
        L.sw_oracle_make_string_range.restype = C.c_int
        L.sw_oracle_score_only.restype = C.c_float
        _LIB = L
    return _LIB


def _scoring(match=3.0, mismatch=-3.0, gap=2.0, lut=None):
    sc = Scoring()
    keep = None
    if lut is not None:
        keep = np.ascontiguousarray(lut, dtype=np.float32).reshape(65536)
        sc.lut = keep.ctypes.data_as(C.POINTER(C.c_float))
    sc.match, sc.mismatch, sc.gap = match, mismatch, gap
    return sc, keep


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1")


def _take(r):
    cx = C.string_at(r.cons_x, r.cons_len).decode("latin-1") if r.cons_len else ""
    cy = C.string_at(r.cons_y, r.cons_len).decode("latin-1") if r.cons_len else ""
    out = dict(score=float(r.score), pos=int(r.pos), end_x=int(r.end_x), end_y=int(r.end_y),
               cons_x=cx, cons_y=cy)
    lib().sw_oracle_free_result(C.byref(r))
    return out


def align(x, y, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
    x, y = _b(x), _b(y)
    sc, keep = _scoring(match, mismatch, gap, lut)
    r = Result()
    rc = lib().sw_oracle_align(x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(sc),
                               C.c_int(semantics), C.byref(r))
    if rc:
        raise MemoryError("sw_oracle_align failed")
    return _take(r)


def align_split(x, y, npiece, overlap_ratio, sm_semantics=F32, la_semantics=F32, match=3.0,
                mismatch=-3.0, gap=2.0, lut=None):
    x, y = _b(x), _b(y)
    sc, keep = _scoring(match, mismatch, gap, lut)
    r = Result()
    piece = C.c_int(0)
    rc = lib().sw_oracle_align_split(x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(sc),
                                     C.c_int(sm_semantics), C.c_int(la_semantics), C.c_int(npiece),
                                     C.c_float(overlap_ratio), C.byref(r), C.byref(piece))
    if rc:
        lib().sw_oracle_free_result(C.byref(r))
        raise AssertionError("reference asserts would fire in _make_string_range")
    out = _take(r)
    out["piece"] = piece.value
    return out


def fill(x, y, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
    """Full matrix as an (m+1, n+1) array (row = index into x, column = index into y)."""
    x, y = _b(x), _b(y)
    m, n = len(x), len(y)
    sc, keep = _scoring(match, mismatch, gap, lut)
    if semantics == F32:
        H = np.zeros((n + 1, m + 1), dtype=np.float32)
        lib().sw_oracle_fill_f32(x, C.c_size_t(m), y, C.c_size_t(n), C.byref(sc), H.ctypes.data_as(C.c_void_p))
    else:
        H = np.zeros((n + 1, m + 1), dtype=np.uint8)
        lib().sw_oracle_fill_u8(x, C.c_size_t(m), y, C.c_size_t(n), C.byref(sc), H.ctypes.data_as(C.c_void_p))
    return H.T


def trace_from(x, y, semantics, start_x, start_y, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
    """Traceback from a given start cell over the matrix of (x, y) (y = window ending at the argmax column);
    result["score"] = H(start) in the window."""
    x, y = _b(x), _b(y)
    sc, keep = _scoring(match, mismatch, gap, lut)
    r = Result()
    L = lib()
    L.sw_oracle_trace_from.restype = C.c_int
    rc = L.sw_oracle_trace_from(x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(sc), C.c_int(semantics),
                                C.c_int64(start_x), C.c_int64(start_y), C.byref(r))
    if rc:
        L.sw_oracle_free_result(C.byref(r))
        raise ValueError("sw_oracle_trace_from: start cell outside the window")
    return _take(r)


def score_only(x, y, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
    x, y = _b(x), _b(y)
    sc, keep = _scoring(match, mismatch, gap, lut)
    return float(lib().sw_oracle_score_only(x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(sc),
                                            C.c_int(semantics)))


def make_string_range(npiece, shortlen, longlen, ratio):
    lefts = (C.c_int64 * max(npiece, 1))()
    rights = (C.c_int64 * max(npiece, 1))()
    rc = lib().sw_oracle_make_string_range(C.c_int(npiece), C.c_int64(shortlen), C.c_int64(longlen),
                                           C.c_float(ratio), lefts, rights)
    if rc:
        return None
    return [(int(lefts[i]), int(rights[i])) for i in range(npiece)]


def true2raw(m, n, ti, tj):
    len_x, len_y = n + 1, m + 1
    nrows, ncols = min(len_x, len_y), max(len_x, len_y)
    ri, rj = C.c_size_t(), C.c_size_t()
    lib().sw_oracle_true2raw(C.c_size_t(ti), C.c_size_t(tj), C.c_size_t(nrows), C.c_size_t(ncols),
                             C.c_size_t(len_x), C.c_size_t(len_y), C.byref(ri), C.byref(rj))
    return ri.value, rj.value


def raw2true(m, n, ri, rj):
    len_x, len_y = n + 1, m + 1
    nrows, ncols = min(len_x, len_y), max(len_x, len_y)
    ti, tj = C.c_size_t(), C.c_size_t()
    lib().sw_oracle_raw2true(C.c_size_t(ri), C.c_size_t(rj), C.c_size_t(nrows), C.c_size_t(ncols),
                             C.c_size_t(len_x), C.c_size_t(len_y), C.byref(ti), C.byref(tj))
    return ti.value, tj.value


def locate(x, y, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
    """(score, index_x, index_y) of find_index_of_maximum without the matrix (sizes beyond memory)."""
    x, y = _b(x), _b(y)
    sc, keep = _scoring(match, mismatch, gap, lut)
    mx, ix, iy = C.c_float(), C.c_int64(), C.c_int64()
    lib().sw_oracle_locate(x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(sc), C.c_int(semantics),
                           C.byref(mx), C.byref(ix), C.byref(iy))
    return mx.value, ix.value, iy.value
