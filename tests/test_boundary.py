"""CPU-side checks of the drop-in boundary: the C ABI library loads and exports every symbol the
header declares, the product never touches the oracle, compute fails loudly without a GPU, and the
host logic of the facade (text store, type inference, token-set builder) behaves like the reference.
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, VOCAB, SHIFT
from vsearch_amd import _native as nat

HEADER = os.path.join(REPO, "include", "vsearch_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"VS_API\s+[\w\s\*]+?\b(vs_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 20
    handle = C.CDLL(nat.LIB_PATH)
    missing = [s for s in syms if not hasattr(handle, s)]
    assert not missing, f"declared in include/vsearch_hip.h but not exported: {missing}"
    assert sorted(nat.EXPORTED_SYMBOLS) == syms, "ctypes signature table and header drifted apart"
    # nothing but the C ABI is visible (-fvisibility=hidden)
    out = subprocess.check_output(["nm", "-D", "--defined-only", nat.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert {s for s in exported if s.startswith("vs_")} == set(syms)


def test_error_slot_and_version():
    lib = nat.lib()
    assert lib.vs_version() >= 100
    rc = lib.vs_index_info(None, None)
    assert rc == nat.VS_EINVAL and "NULL" in nat.last_error()


def test_product_never_imports_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "vsearch_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "vs_oracle" in text.replace("oracle/vs_oracle.c", ""):
                    bad.append(f)
    assert not bad, f"product files reference the oracle: {bad}"
    code = "import sys; import vsearch_amd.ir, vsearch_amd.device_index; assert 'oracle' not in sys.modules"
    subprocess.check_call([sys.executable, "-c", code], cwd=REPO)


def test_compute_fails_loudly_without_gpu(have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    from vsearch_amd.device_index import DeviceIndex
    from vsearch_amd.ir import SparseIndex
    import torch
    with pytest.raises(nat.VsearchNativeError, match="no CPU fallback"):
        DeviceIndex.from_csr(np.array([0, 1]), np.array([0], dtype=np.int32), None, 10)
    idx = SparseIndex()
    idx.vector = torch.eye(4).to_sparse_csr()
    with pytest.raises(nat.VsearchNativeError):
        idx.search(torch.ones(1, 4), 2)
    from vsearch_amd.ir.utils.sparse import build_topk_mask
    with pytest.raises(nat.VsearchNativeError):
        build_topk_mask(torch.rand(2, 8), 3)


def test_missing_library_message(tmp_path, monkeypatch):
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(nat.VsearchNativeError, match="no CPU fallback"):
        nat.lib()


# ---- host logic ------------------------------------------------------------------------------------
def test_get_first_unique_n():
    from vsearch_amd.ir.retriever.index_utils import get_first_unique_n
    assert list(get_first_unique_n([5, 3, 5, 9, 3, 7, 1], 3)) == [5, 3, 9]
    assert list(get_first_unique_n([2, 2, 2], 5)) == [2]
    assert list(get_first_unique_n(iter([]), 4)) == []


def _tokenize(texts, max_len=128):
    out = []
    for t in texts:
        ids = [int(x) for x in str(t).split()]
        out.append(ids[:max_len - 1] + [102] if len(ids) > max_len else ids)
    return out


@pytest.mark.parametrize("tag,max_token,max_len", [("full", 0, 128), ("max16", 16, 128), ("len32", 0, 32)])
def test_bot_build_host_entry_matches_reference_golden(golden, tag, max_token, max_len):
    """vs_bot_build is host-side integer work (tokenisation output -> sorted id sets), so it is checked
    on the CPU against the reference's own CSR (retriever.py:208-253)."""
    g = golden("bot_build")
    toks = _tokenize(g["texts"].tolist(), max_len)
    offsets = np.zeros(len(toks) + 1, np.int64)
    np.cumsum([len(t) for t in toks], out=offsets[1:])
    flat = np.ascontiguousarray(np.concatenate(toks).astype(np.int32))
    indptr = np.empty(len(toks) + 1, np.int64)
    args = (C.c_void_p(flat.ctypes.data), C.c_void_p(offsets.ctypes.data), len(toks), VOCAB, SHIFT, max_token)
    nat.check(nat.lib().vs_bot_build(*args, C.c_void_p(indptr.ctypes.data), None))
    cols = np.empty(int(indptr[-1]), np.int32)
    nat.check(nat.lib().vs_bot_build(*args, C.c_void_p(indptr.ctypes.data), C.c_void_p(cols.ctypes.data)))
    assert (indptr == g[f"{tag}_indptr"]).all() and (cols == g[f"{tag}_indices"]).all()
    bad = flat.copy()
    bad[3] = VOCAB
    with pytest.raises(ValueError, match="out of range"):
        nat.check(nat.lib().vs_bot_build(C.c_void_p(bad.ctypes.data), *args[1:], C.c_void_p(indptr.ctypes.data), None))


def test_text_store_and_low_memory(tmp_path):
    from vsearch_amd.ir import Index, SparseIndex
    docs = ["alpha", "béta ü", {"title": "t", "text": "x"}, "last"]
    p = tmp_path / "corpus.jsonl"
    p.write_text("".join(json.dumps(d) + "\n" for d in docs), encoding="utf-8")
    a = Index(None, str(p))
    assert len(a) == 4 and a.get_sample(1) == "béta ü" and a.data[2]["title"] == "t"
    b = SparseIndex(None, str(p), low_memory=True)          # broken upstream (index.py:51-52,68-86); works here
    assert b.data is None and [b.get_sample(i) for i in range(4)] == docs and len(b) == 4
    assert len(Index()) == 0
    s = str(a)
    assert s.splitlines()[0].replace(" ", "") == "IndexType:Index" and "Number of Texts   : 4" in s and "Device" in s


def test_index_type_and_results_types():
    from vsearch_amd.ir import IndexType, SearchResults, Index, SparseIndex, BoTIndex
    assert [t.value for t in IndexType] == ["dense", "sparse", "bag_of_token"]
    assert IndexType("bag_of_token") is IndexType.BAG_OF_TOKEN
    ids, scores = SearchResults([1], [2.0])                  # callers unpack positionally (retriever.py:182)
    assert ids == [1] and scores == [2.0]
    assert (Index.index_type, SparseIndex.index_type, BoTIndex.index_type) == (IndexType.DENSE, IndexType.SPARSE, IndexType.BAG_OF_TOKEN)
    assert issubclass(BoTIndex, SparseIndex) and issubclass(SparseIndex, Index)


def test_load_index_type_inference_errors():
    from vsearch_amd.ir import Retriever
    r = Retriever.__new__(Retriever)
    with pytest.raises(ValueError, match="Cannot infer"):
        Retriever.load_index(r, index_file="x.bin")
    with pytest.raises(TypeError):
        Retriever.load_index(r, index_file="x.npz", index_type=3)
    with pytest.raises(ValueError):
        Retriever.load_index(r, index_file="x.npz", index_type="inverted")
    with pytest.raises(TypeError):
        Retriever.build_index(r, ["a"], index_type=1.5)


def test_src_ir_import_shim():
    from src.ir import Retriever, RetrieverConfig       # noqa: F401  (test/quick_start.py:2)
    from src.ir.retriever.index import SparseIndex       # noqa: F401
    from src.ir.utils.sparse import build_bow_mask       # noqa: F401
    import vsearch_amd.ir as ir
    assert Retriever is ir.Retriever


@pytest.mark.parametrize("gen,args,header", [("gen_quad_asm.py", ["4"], "bp_quad_asm.h"), ("gen_bq_asm.py", ["8"], "bp_bq_asm.h"), ("gen_head_asm.py", [], "bp_head_asm.h")])
def test_generated_asm_headers_match_their_generators(tmp_path, gen, args, header):
    """The walks' inner loops and the head product are generated inline-asm statements (tools/gen_*_asm.py -> csrc/*_asm.h): the committed
    header is what the committed generator writes (its first line names the output path)."""
    out = tmp_path / header
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", gen)] + args + [str(out)], cwd=REPO, stdout=subprocess.DEVNULL)
    want = open(os.path.join(REPO, "vsearch_amd", "csrc", header)).read().split("\n", 1)[1]
    assert out.read_text().split("\n", 1)[1] == want, f"{header} is not what tools/{gen} {' '.join(args)} generates"
