"""parallel-genomeseq_amd — MI355X-native Smith-Waterman engine behind the LocalAligner /
ParallelLocalAligner API of kosta777/parallel-genomeseq.  See DESIGN.md.

The compute path is the HIP library parallel-genomeseq_amd/libmi355_sw.so (C-ABI in
include/mi355_sw.h).  Importing this package does not need a GPU; running an alignment does, and
fails loudly without one — there is no CPU fallback."""
from . import synth  # noqa: F401
from . import capi  # noqa: F401
from .aligner import (LocalAligner, OMPParallelLocalAligner, ParallelLocalAligner, SWAligner,  # noqa: F401
                      Similarity_Matrix, Similarity_Matrix_Skewed, default_context)
from .capi import F32, U8SAT, Context, MI355Error, MultiContext  # noqa: F401
