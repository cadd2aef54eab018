"""Drop-in mirror of the reference's ``src.ir`` package for the vocabulary-space retrieval path
(/root/reference/src/ir/__init__.py:1): ``from vsearch_amd.ir import Retriever, RetrieverConfig``."""
from .retriever.retriever import Retriever, RetrieverConfig
from .retriever.index import BoTIndex, Index, IndexType, SearchResults, SparseIndex

__all__ = ["Retriever", "RetrieverConfig", "Index", "SparseIndex", "BoTIndex", "IndexType", "SearchResults"]
