"""Import-path shim: lets scripts written for the reference (``from src.ir import Retriever``,
``from src.ir.retriever.index import SparseIndex``) run on vsearch_amd unchanged."""
import importlib
import sys

from vsearch_amd.ir import *  # noqa: F401,F403
from vsearch_amd.ir import __all__  # noqa: F401

for _sub in ("retriever", "retriever.index", "retriever.retriever", "retriever.index_utils", "encoder", "encoder.vdr",
             "encoder.types", "biencoder", "biencoder.biencoder", "utils", "utils.sparse"):
    sys.modules[f"{__name__}.{_sub}"] = importlib.import_module(f"vsearch_amd.ir.{_sub}")
