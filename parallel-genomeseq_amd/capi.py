"""ctypes binding of libmi355_sw.so (include/mi355_sw.h).  There is no CPU fallback: if the HIP
library is missing, or no MI355X is visible, every compute call raises."""
import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355_sw.so")

F32, U8SAT = 0, 1
SCORE_ONLY = 1

EXPORTS = [
    "mi355_sw_create", "mi355_sw_destroy", "mi355_sw_last_error", "mi355_sw_default_params",
    "mi355_sw_align", "mi355_sw_set_reference", "mi355_sw_align_batch", "mi355_sw_batch_upload",
    "mi355_sw_batch_run", "mi355_sw_score_ranges", "mi355_sw_align_scored_range", "mi355_sw_align_split", "mi355_sw_make_string_range", "mi355_sw_fill_matrix",
    "mi355_sw_argmax", "mi355_sw_true2raw", "mi355_sw_raw2true", "mi355_sw_last_timings", "mi355_sw_free_result", "mi355_sw_free_results",
    "mi355_sw_build_info", "mi355_sw_last_kernel", "mi355_sw_batch_run_view",
    "mi355_sw_multi_create", "mi355_sw_multi_destroy", "mi355_sw_multi_last_error", "mi355_sw_multi_device_count",
    "mi355_sw_multi_rccl_version", "mi355_sw_multi_align_split", "mi355_sw_multi_set_reference",
    "mi355_sw_multi_align_batch", "mi355_sw_multi_last_timings",
    "mi355_sw_set_option", "mi355_sw_option_names", "mi355_sw_multi_set_option", "mi355_sw_last_counters", "mi355_sw_last_counter", "mi355_sw_batch_upload_packed", "mi355_sw_best_range", "mi355_sw_last_path",
]
MULTI_RCCL = 1


class Params(C.Structure):
    _fields_ = [("lut", C.POINTER(C.c_float)), ("match", C.c_float), ("mismatch", C.c_float),
                ("gap", C.c_float), ("semantics", C.c_int)]


class Result(C.Structure):
    _fields_ = [("score", C.c_float), ("pos", C.c_uint32), ("end_x", C.c_int64), ("end_y", C.c_int64),
                ("cons_x", C.c_void_p), ("cons_y", C.c_void_p), ("cons_len", C.c_size_t),
                ("timings_us", C.c_float * 2)]


class BatchView(C.Structure):
    _fields_ = [("n", C.c_size_t), ("score", C.POINTER(C.c_float)), ("pos", C.POINTER(C.c_uint32)),
                ("end_x", C.POINTER(C.c_int64)), ("end_y", C.POINTER(C.c_int64)), ("cons_x", C.POINTER(C.c_void_p)),
                ("cons_y", C.POINTER(C.c_void_p)), ("cons_len", C.POINTER(C.c_uint32)), ("timings_us", C.c_float * 2)]


class KernelInfo(C.Structure):
    _fields_ = [("cell", C.c_int), ("lanes", C.c_int), ("rows_per_lane", C.c_int), ("strips", C.c_int), ("twin", C.c_int),
                ("chunk_len", C.c_int64), ("sub_len", C.c_int64), ("warm", C.c_int64), ("cells", C.c_double),
                ("valu_ops_per_cell", C.c_double), ("name", C.c_char * 160)]


CELL_NAMES = {0: "i16", 1: "u8", 2: "f32", 3: "u8", 4: "f16", 5: "u8"}      # arithmetic type of the cells ("dtype" of bench.py)

_RESULT_DTYPE = np.dtype({"names": ["score", "pos", "end_x", "end_y", "cons_x", "cons_y", "cons_len", "t0", "t1"],
                          "formats": ["<f4", "<u4", "<i8", "<i8", "<u8", "<u8", "<u8", "<f4", "<f4"],
                          "offsets": [Result.score.offset, Result.pos.offset, Result.end_x.offset, Result.end_y.offset,
                                      Result.cons_x.offset, Result.cons_y.offset, Result.cons_len.offset,
                                      Result.timings_us.offset, Result.timings_us.offset + 4],
                          "itemsize": C.sizeof(Result)})


class MI355Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mi355_sw error %d: %s" % (code, msg))
        self.code = code


_LIB = None


def lib():
    """Load the HIP library; fail loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmi355_sw.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C parallel-genomeseq_amd/csrc`")
        L = C.CDLL(LIB_PATH)
        L.mi355_sw_last_error.restype = C.c_char_p
        L.mi355_sw_build_info.restype = C.c_char_p
        L.mi355_sw_multi_last_error.restype = C.c_char_p
        L.mi355_sw_option_names.restype = C.c_char_p
        L.mi355_sw_last_path.restype = C.c_char_p
        for name in EXPORTS:
            getattr(L, name)
        _LIB = L
    return _LIB


def make_params(semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
    p = Params()
    keep = None
    if lut is not None:
        keep = np.ascontiguousarray(lut, dtype=np.float32).reshape(65536)
        p.lut = keep.ctypes.data_as(C.POINTER(C.c_float))
    p.match, p.mismatch, p.gap, p.semantics = match, mismatch, gap, semantics
    return p, keep


def _bytes(s):
    if isinstance(s, (bytes, bytearray)):
        return bytes(s)
    if isinstance(s, np.ndarray):
        return s.astype(np.uint8).tobytes()
    return s.encode("latin-1")


def _take(r):
    cx = C.string_at(r.cons_x, r.cons_len).decode("latin-1") if r.cons_len else ""
    cy = C.string_at(r.cons_y, r.cons_len).decode("latin-1") if r.cons_len else ""
    return dict(score=float(r.score), pos=int(r.pos), end_x=int(r.end_x), end_y=int(r.end_y),
                cons_x=cx, cons_y=cy, timings_us=(float(r.timings_us[0]), float(r.timings_us[1])))


class Context:
    """One engine context = one GPU (mi355_sw_create)."""

    def __init__(self, device=0):
        self._L = lib()
        self._ctx = C.c_void_p()
        rc = self._L.mi355_sw_create(C.byref(self._ctx), C.c_int(device))
        if rc:
            raise MI355Error(rc, "mi355_sw_create failed (no usable HIP device %d?)" % device)
        self.device = device
        self._view = None              # struct-of-arrays view of the last batch_run(raw=True): library memory, see consensus()
        self._nbatch = 0
        self.last_call_s = 0.0         # wall time of the last mi355_sw_batch_run_view call itself (batch_run(raw=True))

    def close(self):
        self._view = None
        if self._ctx:
            self._L.mi355_sw_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        self._view = None              # every library call invalidates the memory the last view points into
        if rc:
            raise MI355Error(rc, (self._L.mi355_sw_last_error(self._ctx) or b"").decode())

    def set_option(self, key, value=True):
        """A/B and diagnostic switches of this context (mi355_sw_set_option); none changes a result."""
        v = None if value in (None, False, 0) else ("1" if value is True else str(value))
        self._chk(self._L.mi355_sw_set_option(self._ctx, key.encode(), v.encode() if v is not None else None))

    # -- single alignment ---------------------------------------------------------------------
    def align(self, x, y, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
        x, y = _bytes(x), _bytes(y)
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        r = Result()
        self._chk(self._L.mi355_sw_align(self._ctx, x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(p), C.byref(r)))
        out = _take(r)
        self._L.mi355_sw_free_result(C.byref(r))
        return out

    def argmax(self, x, y, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
        x, y = _bytes(x), _bytes(y)
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        ix, iy, mx = C.c_int64(), C.c_int64(), C.c_float()
        self._chk(self._L.mi355_sw_argmax(self._ctx, x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(p),
                                          C.byref(ix), C.byref(iy), C.byref(mx)))
        return ix.value, iy.value, mx.value

    def fill_matrix(self, x, y, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
        x, y = _bytes(x), _bytes(y)
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        H = np.zeros((len(y) + 1, len(x) + 1), dtype=np.float32)
        self._chk(self._L.mi355_sw_fill_matrix(self._ctx, x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(p),
                                               H.ctypes.data_as(C.POINTER(C.c_float))))
        return H.T

    def align_split(self, x, y, npiece, overlap_ratio, sm_semantics=F32, la_semantics=F32, match=3.0,
                    mismatch=-3.0, gap=2.0, lut=None):
        x, y = _bytes(x), _bytes(y)
        p, keep = make_params(sm_semantics, match, mismatch, gap, lut)
        r = Result()
        piece = C.c_int(0)
        self._chk(self._L.mi355_sw_align_split(self._ctx, x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(p),
                                               C.c_int(sm_semantics), C.c_int(la_semantics), C.c_int(npiece),
                                               C.c_float(overlap_ratio), C.byref(r), C.byref(piece)))
        out = _take(r)
        out["piece"] = piece.value
        self._L.mi355_sw_free_result(C.byref(r))
        return out

    # -- batches ------------------------------------------------------------------------------
    def set_reference(self, y):
        y = _bytes(y)
        self._chk(self._L.mi355_sw_set_reference(self._ctx, y, C.c_size_t(len(y))))
        self._ref_len = len(y)

    def batch_upload(self, xs):
        xs = [_bytes(x) for x in xs]
        n = len(xs)
        arr = (C.c_char_p * n)(*xs)
        lens = (C.c_size_t * n)(*[len(x) for x in xs])
        self._chk(self._L.mi355_sw_batch_upload(self._ctx, C.c_size_t(n), arr, lens))
        self._nbatch = n

    def batch_upload_packed(self, buf, offsets):
        """The batch as one contiguous buffer (bytes or uint8 array) and n + 1 ascending int64 offsets
        (mi355_sw_batch_upload_packed): no per-sequence Python objects or pointers."""
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offs) - 1
        if isinstance(buf, np.ndarray):
            arr = np.ascontiguousarray(buf, dtype=np.uint8)
            ptr = arr.ctypes.data_as(C.c_char_p)
        else:
            arr = bytes(buf)
            ptr = C.c_char_p(arr)
        self._chk(self._L.mi355_sw_batch_upload_packed(self._ctx, C.c_size_t(n), ptr, offs.ctypes.data_as(C.POINTER(C.c_int64))))
        self._nbatch = n

    def batch_run(self, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None, flags=0, raw=False):
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        n = self._nbatch
        if raw:
            # struct-of-arrays view in library-owned memory (mi355_sw_batch_run_view): no per-result objects, no
            # per-result allocations on either side; cons=True adds the strings' lengths and addresses
            v = BatchView()
            t0 = time.perf_counter()
            rc = self._L.mi355_sw_batch_run_view(self._ctx, C.byref(p), C.c_int(flags), C.byref(v))
            self.last_call_s = time.perf_counter() - t0            # the C-ABI call alone (what a C / C++ caller waits for)
            self._chk(rc)
            if n == 0:
                return dict(score=np.zeros(0, np.float32), pos=np.zeros(0, np.int64), end_x=np.zeros(0, np.int64),
                            end_y=np.zeros(0, np.int64), cons_len=np.zeros(0, np.int64))
            arr = lambda ptr, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt)   # copies out of library memory
            out = dict(score=arr(v.score, np.float32), pos=arr(v.pos, np.int64), end_x=arr(v.end_x, np.int64),
                       end_y=arr(v.end_y, np.int64), cons_len=arr(v.cons_len, np.int64))
            self._view = v                                             # consensus(k) reads through it until the next call
            return out
        res = (Result * n)()
        self._chk(self._L.mi355_sw_batch_run(self._ctx, C.byref(p), C.c_int(flags), res))
        out = [_take(r) for r in res]
        self._L.mi355_sw_free_results(res, C.c_size_t(n))
        return out

    def consensus(self, k):
        """(cons_x, cons_y) of alignment k of the last batch_run(raw=True) (valid until the next call)."""
        v = self._view
        if v is None:
            raise RuntimeError("consensus(k): no batch_run(raw=True) result is current (the view dies with the next call on "
                               "this context, with an empty batch and with close())")
        if not 0 <= k < int(v.n):
            raise IndexError("consensus(%d): the batch has %d alignments" % (k, int(v.n)))
        ln = int(v.cons_len[k])
        if ln == 0:
            return "", ""
        return C.string_at(v.cons_x[k], ln).decode("latin-1"), C.string_at(v.cons_y[k], ln).decode("latin-1")

    def score_ranges(self, ranges, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None):
        """Per-range maxima of every resident query: array [len(ranges), n_queries]."""
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        n = len(ranges)
        lefts = (C.c_int64 * n)(*[r[0] for r in ranges])
        rights = (C.c_int64 * n)(*[r[1] for r in ranges])
        out = np.zeros((n, self._nbatch), dtype=np.float32)
        self._chk(self._L.mi355_sw_score_ranges(self._ctx, C.c_size_t(n), lefts, rights, C.byref(p),
                                                out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def best_range(self, ranges, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None, known_best=0.0, want_exact_above=False):
        """Per resident query the first range with the greatest maximum (mi355_sw_best_range): (best[nq], best_range[nq],
        maxima[len(ranges), nq] — exact for the winners, lower bounds for ranges that cannot win).  want_exact_above=True (the
        multi-rank protocol): a fourth value, the maximum above which this call was exact (-1: everything); the caller merges
        the ranks' bests and calls again with known_best = the merged best when that does not exceed it."""
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        n = len(ranges)
        lefts = (C.c_int64 * max(1, n))(*[r[0] for r in ranges])
        rights = (C.c_int64 * max(1, n))(*[r[1] for r in ranges])
        mx = np.zeros((n, self._nbatch), dtype=np.float32)
        best = np.zeros(self._nbatch, dtype=np.float32)
        which = np.zeros(self._nbatch, dtype=np.int64)
        above = C.c_float(-1.0)
        self._chk(self._L.mi355_sw_best_range(self._ctx, C.c_size_t(n), lefts, rights, C.byref(p), C.c_float(known_best),
                                              mx.ctypes.data_as(C.POINTER(C.c_float)), best.ctypes.data_as(C.POINTER(C.c_float)),
                                              which.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(above) if want_exact_above else None))
        if want_exact_above:
            return best, which, mx, float(above.value)
        return best, which, mx

    def align_scored_range(self, k, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None, flags=0, query=0):
        """Range k of the last score_ranges call finished as a stand-alone problem (argmax + traceback from the sweep's
        keys when the scoring is the same); result of resident query `query`, pos relative to the range start."""
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        n = self._nbatch
        res = (Result * max(1, n))()
        self._chk(self._L.mi355_sw_align_scored_range(self._ctx, C.c_size_t(k), C.byref(p), C.c_int(flags), res))
        out = _take(res[query])
        self._L.mi355_sw_free_results(res, C.c_size_t(n))
        return out

    def align_batch(self, xs, y=None, **kw):
        if y is not None:
            self.set_reference(y)
        self.batch_upload(xs)
        return self.batch_run(**kw)

    def last_timings(self):
        t = (C.c_double * 6)()
        self._L.mi355_sw_last_timings(self._ctx, t)
        return dict(score_us=t[0], locate_us=t[1], trace_us=t[2], total_us=t[3], score_launches=int(t[4]), cells=t[5])


    def last_counters(self):
        """Candidate-filter counters of the last call (mi355_sw_last_counters)."""
        c = (C.c_uint64 * 4)()
        self._L.mi355_sw_last_counters(self._ctx, c)
        out = dict(requeried=int(c[0]), whole_batch_again=int(c[1]), candidates=int(c[2]), left_window=int(c[3]))
        for name in ("first_settled", "saved_locates", "saved_traces", "saved_fallbacks", "wait_retries", "early_settled", "beyond_f16"):
            v = C.c_uint64(0)
            rc = self._L.mi355_sw_last_counter(self._ctx, name.encode(), C.byref(v))
            if rc:
                raise MI355Error(rc, "mi355_sw_last_counter(%s)" % name)
            out[name] = int(v.value)
        return out

    def last_path(self):
        """Tags of the kernels / pipeline decisions of the last call (mi355_sw_last_path), as a list."""
        return self._L.mi355_sw_last_path(self._ctx).decode().split()

    def last_kernel(self):
        """The sw_score_kernel instance that swept the most cells in the last call (mi355_sw_last_kernel)."""
        k = KernelInfo()
        self._L.mi355_sw_last_kernel(self._ctx, C.byref(k))
        return dict(cell=k.cell, dtype=CELL_NAMES.get(k.cell, "?"), lanes=k.lanes, rows_per_lane=k.rows_per_lane,
                    strips=bool(k.strips), twin=bool(k.twin), chunk_len=k.chunk_len, sub_len=k.sub_len, warm=k.warm,
                    cells=k.cells, valu_ops_per_cell=k.valu_ops_per_cell, name=k.name.decode())


class MultiContext:
    """Several GPUs of one node behind one handle (mi355_sw_multi_*): one engine context and one host thread per
    device inside this process.  devices=None: all visible devices; a device may be listed twice (rehearsal on a
    one-GPU box) unless rccl=True."""

    def __init__(self, devices=None, rccl=False):
        self._L = lib()
        self._m = C.c_void_p()
        n = 0 if devices is None else len(devices)
        arr = (C.c_int * max(1, n))(*(devices or [0]))
        rc = self._L.mi355_sw_multi_create(C.byref(self._m), C.c_int(n), arr if n else None, C.c_int(MULTI_RCCL if rccl else 0))
        if rc:
            raise MI355Error(rc, "mi355_sw_multi_create failed (devices %r, rccl=%r)" % (devices, rccl))
        self.ndev = self._L.mi355_sw_multi_device_count(self._m)
        self.rccl_version = self._L.mi355_sw_multi_rccl_version(self._m)

    def close(self):
        if self._m:
            self._L.mi355_sw_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise MI355Error(rc, (self._L.mi355_sw_multi_last_error(self._m) or b"").decode())

    def set_option(self, key, value=True):
        v = None if value in (None, False, 0) else ("1" if value is True else str(value))
        self._chk(self._L.mi355_sw_multi_set_option(self._m, key.encode(), v.encode() if v is not None else None))

    def align_split(self, x, y, npiece, overlap_ratio, sm_semantics=F32, la_semantics=F32, match=3.0,
                    mismatch=-3.0, gap=2.0, lut=None):
        x, y = _bytes(x), _bytes(y)
        p, keep = make_params(sm_semantics, match, mismatch, gap, lut)
        r = Result()
        piece = C.c_int(0)
        self._chk(self._L.mi355_sw_multi_align_split(self._m, x, C.c_size_t(len(x)), y, C.c_size_t(len(y)), C.byref(p),
                                                     C.c_int(sm_semantics), C.c_int(la_semantics), C.c_int(npiece),
                                                     C.c_float(overlap_ratio), C.byref(r), C.byref(piece)))
        out = _take(r)
        out["piece"] = piece.value
        self._L.mi355_sw_free_result(C.byref(r))
        return out

    def set_reference(self, y):
        y = _bytes(y)
        self._chk(self._L.mi355_sw_multi_set_reference(self._m, y, C.c_size_t(len(y))))

    def align_batch(self, xs, y=None, semantics=F32, match=3.0, mismatch=-3.0, gap=2.0, lut=None, flags=0, raw=False):
        """Returns (results, best_index)."""
        if y is not None:
            self.set_reference(y)
        xs = [_bytes(x) for x in xs]
        n = len(xs)
        arr = (C.c_char_p * max(1, n))(*xs)
        lens = (C.c_size_t * max(1, n))(*[len(x) for x in xs])
        p, keep = make_params(semantics, match, mismatch, gap, lut)
        res = (Result * max(1, n))()
        best = C.c_int64(-1)
        self._chk(self._L.mi355_sw_multi_align_batch(self._m, C.c_size_t(n), arr, lens, C.byref(p), C.c_int(flags), res,
                                                     C.byref(best)))
        if raw:
            v = np.frombuffer(res, dtype=_RESULT_DTYPE, count=n)
            out = dict(score=v["score"].astype(np.float32), pos=v["pos"].astype(np.int64),
                       end_x=v["end_x"].astype(np.int64), end_y=v["end_y"].astype(np.int64))
        else:
            out = [_take(res[k]) for k in range(n)]
        self._L.mi355_sw_free_results(res, C.c_size_t(n))
        return out, best.value

    def last_timings(self):
        t = (C.c_double * 6)()
        self._L.mi355_sw_multi_last_timings(self._m, t)
        return dict(score_us=t[0], locate_us=t[1], trace_us=t[2], total_us=t[3], score_launches=int(t[4]), cells=t[5])


def option_names():
    return lib().mi355_sw_option_names().decode().split(",")


def make_string_range(npiece, shortlen, longlen, ratio):
    lefts = (C.c_int64 * max(1, npiece))()
    rights = (C.c_int64 * max(1, npiece))()
    rc = lib().mi355_sw_make_string_range(C.c_int(npiece), C.c_int64(shortlen), C.c_int64(longlen), C.c_float(ratio),
                                          lefts, rights)
    if rc:
        return None
    return [(int(lefts[k]), int(rights[k])) for k in range(npiece)]


def true2raw(nx, ny, ti, tj):
    ri, rj = C.c_size_t(), C.c_size_t()
    lib().mi355_sw_true2raw(C.c_size_t(nx), C.c_size_t(ny), C.c_size_t(ti), C.c_size_t(tj), C.byref(ri), C.byref(rj))
    return ri.value, rj.value


def raw2true(nx, ny, ri, rj):
    ti, tj = C.c_size_t(), C.c_size_t()
    lib().mi355_sw_raw2true(C.c_size_t(nx), C.c_size_t(ny), C.c_size_t(ri), C.c_size_t(rj), C.byref(ti), C.byref(tj))
    return ti.value, tj.value
