"""parallel-genomeseq_amd — MI355X-native Smith-Waterman engine behind the LocalAligner /
ParallelLocalAligner API of kosta777/parallel-genomeseq.  See DESIGN.md."""
from . import synth  # noqa: F401
