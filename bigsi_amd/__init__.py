"""bigsi_amd: the BIGSI query path on AMD MI355X (gfx950) -- `from bigsi_amd import BIGSI`.

Mirrors the import surface of the reference package (bigsi/__init__.py:1-3) for the hot path: the index object,
its storage registry, and the k-mer helpers.  See DESIGN.md for what runs where."""
from .version import __version__
from .utils import seq_to_kmers, reverse_comp, canonical, convert_query_kmer, convert_query_kmers
from .bitrow import BitRow
from .graph.bigsi import BIGSI, BigsiQueryResult

__all__ = ["BIGSI", "BigsiQueryResult", "BitRow", "seq_to_kmers", "reverse_comp", "canonical",
           "convert_query_kmer", "convert_query_kmers", "__version__"]
