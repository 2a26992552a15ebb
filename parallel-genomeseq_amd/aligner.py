"""Host-side mirror of the reference's aligner interface (src/aligner/localaligner.h:7-28,
smithwaterman.h:11-58, plocalaligner.h:6-33) on top of the C-ABI: same class and method names,
same argument meaning, same defaults, same error behaviour where the reference defines one.

    la = SWAligner(first, second)                 # Similarity_Matrix (float32) semantics
    la = SWAligner(first, second, matrix=Similarity_Matrix_Skewed)
    la.calculateScore(); la.getScore(); la.getPos(); la.getConsensus_x(); la.getConsensus_y()

`first` = rows (the read), `second` = columns (the reference); pos indexes `second`, 1-based.
"""
import numpy as np

from . import capi


class Similarity_Matrix:            # tag types mirroring similaritymatrix.h:26-62
    semantics = capi.F32


class Similarity_Matrix_Skewed:     # similaritymatrix.h:64-100
    semantics = capi.U8SAT


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = capi.Context(device)
    return _default_ctx[device]


def _lut_from_function(fn):
    lut = np.empty((256, 256), dtype=np.float32)
    for a in range(256):
        ca = chr(a)
        for b in range(256):
            lut[a, b] = fn(ca, chr(b))
    return lut


class LocalAligner:
    """localaligner.h:7-17"""

    def calculateScore(self): raise NotImplementedError
    def getScore(self): raise NotImplementedError
    def getPos(self): raise NotImplementedError
    def getConsensus_x(self): raise NotImplementedError
    def getConsensus_y(self): raise NotImplementedError
    def getTimings(self): raise NotImplementedError


class SWAligner(LocalAligner):
    """smithwaterman.h:11-58.  `scoring` is either None (3 / -3), a callable f(a, b) -> float (tabulated
    once into a 256x256 table, the way the GPU consumes std::function scoring), or a 256x256 array."""

    def __init__(self, first_sequence, second_sequence, scoring=None, gap_penalty=2.0, matrix=Similarity_Matrix,
                 context=None):
        self.sequence_x, self.sequence_y = first_sequence, second_sequence
        self.gap_penalty = float(gap_penalty)
        self.matrix = matrix
        self._lut = None
        if scoring is not None:
            self._lut = _lut_from_function(scoring) if callable(scoring) else np.asarray(scoring, dtype=np.float32)
        self._ctx = context
        self.pos, self.max_score = 0, -1.0          # smithwaterman.cpp:27-28
        self.consensus_x, self.consensus_y = "", ""
        self._timings = (0.0, 0.0)
        self._end = (0, 0)

    def _context(self):
        return self._ctx if self._ctx is not None else default_context()

    def calculateScore(self):
        r = self._context().align(self.sequence_x, self.sequence_y, self.matrix.semantics, gap=self.gap_penalty,
                                  lut=self._lut)
        self.max_score, self.pos = r["score"], r["pos"]
        self.consensus_x, self.consensus_y = r["cons_x"], r["cons_y"]
        self._timings, self._end = r["timings_us"], (r["end_x"], r["end_y"])
        return self.max_score

    def getScore(self): return self.max_score
    def getPos(self): return self.pos
    def getConsensus_x(self): return self.consensus_x
    def getConsensus_y(self): return self.consensus_y
    def getTimings(self): return self._timings

    def getSimilarity_matrix(self):
        """Matrix accessor (operator()(row, col)): the full matrix recomputed on the device."""
        return self._context().fill_matrix(self.sequence_x, self.sequence_y, self.matrix.semantics,
                                           gap=self.gap_penalty, lut=self._lut)

    def find_index_of_maximum(self):
        return self._context().argmax(self.sequence_x, self.sequence_y, self.matrix.semantics, gap=self.gap_penalty,
                                      lut=self._lut)


class ParallelLocalAligner:
    """localaligner.h:19-28"""


class OMPParallelLocalAligner(ParallelLocalAligner):
    """plocalaligner.h:6-33: reference split into `npiece` overlapping pieces (overlap =
    overlap_ratio * |first|); the winning piece is re-aligned with DEFAULT scoring by `aligner_matrix`."""

    def __init__(self, first_sequence, second_sequence, npiece, overlap_ratio, scoring=None, gap_penalty=2.0,
                 matrix=Similarity_Matrix, aligner_matrix=None, context=None):
        self.sequence_x, self.sequence_y = first_sequence, second_sequence
        self.npiece, self.overlap_ratio = int(npiece), float(overlap_ratio)
        self.gap_penalty = float(gap_penalty)
        self.matrix = matrix
        self.aligner_matrix = aligner_matrix if aligner_matrix is not None else matrix
        self._lut = None
        if scoring is not None:
            self._lut = _lut_from_function(scoring) if callable(scoring) else np.asarray(scoring, dtype=np.float32)
        self._ctx = context
        self.pos, self.max_score = 0, -1.0          # plocalaligner.cpp:78-79
        self.consensus_x, self.consensus_y = "", ""
        self._timings = (0.0, 0.0)
        self.winning_piece = 0
        # the reference asserts in its constructor (plocalaligner.cpp:52,63,65)
        if capi.make_string_range(self.npiece, len(first_sequence), len(second_sequence), self.overlap_ratio) is None:
            raise AssertionError("_make_string_range: overlaplength <= piecelength / right < longstringlength")

    def calculateScore(self):
        ctx = self._ctx if self._ctx is not None else default_context()
        r = ctx.align_split(self.sequence_x, self.sequence_y, self.npiece, self.overlap_ratio,
                            self.matrix.semantics, self.aligner_matrix.semantics, gap=self.gap_penalty, lut=self._lut)
        self.max_score, self.pos = r["score"], r["pos"]
        self.consensus_x, self.consensus_y = r["cons_x"], r["cons_y"]
        self._timings, self.winning_piece = r["timings_us"], r["piece"]
        return self.max_score

    def getScore(self): return self.max_score
    def getPos(self): return self.pos
    def getConsensus_x(self): return self.consensus_x
    def getConsensus_y(self): return self.consensus_y
    def getTimings(self): return self._timings
