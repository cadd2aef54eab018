"""vsearch_amd -- MI355X-native implementation of vsearch's vocabulary-space retrieval hot path.

    from vsearch_amd.ir import Retriever, SparseIndex       # mirror of the reference's src.ir
    from vsearch_amd.device_index import DeviceIndex          # thin handle over the C ABI

Compute lives in ``libvsearch_hip.so`` (hand-written HIP for gfx950, C ABI in include/vsearch_hip.h);
importing this package does not load it -- the first compute call does, and raises if the library or
a GPU is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
