"""Text-to-text quick start on one MI355X (the flow of the reference's test/quick_start.py and README.md:99-140).

    python examples/quick_start.py [checkpoint]

`checkpoint` is a local directory written by `Retriever.save_pretrained` / downloaded from the hub (vsearch/vdr-nq,
vsearch/svdr-nq), or `random:<hidden>:<layers>:<seed>` for a randomly initialised encoder with the id tokenizer
(no files needed: the plumbing is the same, the scores are meaningless)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a source checkout

from vsearch_amd.inference.common import load_retriever
from vsearch_amd.ir.retriever.index import IndexType

checkpoint = sys.argv[1] if len(sys.argv) > 1 else "random:64:2:0"
random_init = checkpoint.startswith("random:")
retriever = load_retriever(checkpoint, "cuda")

if random_init:   # the id tokenizer reads whitespace-separated token ids
    query = "101 2054 2024 1996 6666 1997 5948 2665 5572 102"
    passages = ["101 2665 5572 2003 2124 2005 6666 102", "101 1996 2381 1997 4157 5246 2067 102", "101 13272 2003 1037 2568 102"]
else:
    query = "What are the benefits of drinking green tea?"
    passages = ["Green tea is known for its antioxidant properties, which can help protect cells from damage caused by free radicals.",
                "The history of coffee dates back to ancient times, with its origins in Ethiopia.",
                "Yoga is a mind-body practice that combines physical postures, breathing exercises, and meditation."]

# 1. embeddings in vocabulary space: [N, 29523] with topk non-zeros (+ the tokens present in the text)
q_emb = retriever.encoder_q.embed(query, topk=768)
p_emb = retriever.encoder_p.embed(passages, topk=768)
print("relevance:", (q_emb @ p_emb.t()).tolist())

# 2. index + search (SparseIndex: CSR store in HBM, scored by the HIP scan kernels)
retriever.build_index(passages, index_type=IndexType.SPARSE)
ids, scores = retriever.retrieve(query, k=min(3, len(passages)))
print("top ids:", ids.tolist(), "scores:", scores.tolist())
print("best passage:", retriever.index.get_sample(int(ids[0, 0])))

# 3. bag-of-token index (SVDR beta search) with rerank
retriever.build_index(passages, index_type=IndexType.BAG_OF_TOKEN)
ids, scores = retriever.retrieve(query, k=min(3, len(passages)), rerank=True)
print("beta search + rerank:", ids.tolist())
torch.cuda.synchronize()
