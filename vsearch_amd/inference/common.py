"""Shared plumbing of the inference CLIs.  Upstream calls `python -m inference.build_index.*`,
`inference.search.*`, `inference.score.eval_wiki21m` (examples/inference_sparse/README.md:71-141,
test/svdr_wiki21m/*.sh) but that package is absent from the reference snapshot; these modules are its
counterparts with the documented flags, on top of vsearch_amd.ir."""
from __future__ import annotations

import json
import logging
import time
from typing import List

import torch

logger = logging.getLogger("vsearch_amd.inference")


def read_jsonl(path: str) -> List:
    with open(path, "r", encoding="utf-8") as fh:
        return [json.loads(line) for line in fh if line.strip()]


def shard_slice(n: int, num_shard: int, shard_id: int):
    """Contiguous shard `shard_id` of `num_shard` (README.md:90-107)."""
    per = -(-n // num_shard)
    return slice(min(n, shard_id * per), min(n, (shard_id + 1) * per))


class WhitespaceIdTokenizer:
    """Offline stand-in used by `--checkpoint random:...`: texts are space-separated token ids."""

    def __init__(self, vocab_size=30522):
        self.vocab = range(vocab_size)

    def _ids(self, texts, max_length, truncation):
        out = []
        for t in texts:
            ids = [int(x) for x in str(t).split()]
            if truncation and max_length and len(ids) > max_length:
                ids = ids[:max_length - 1] + [102]
            out.append(ids)
        return out

    def __call__(self, texts, max_length=None, truncation=False):
        return {"input_ids": self._ids(texts, max_length, truncation)}

    def batch_encode_plus(self, texts, padding=True, truncation=True, max_length=None, return_tensors="pt"):
        from transformers import BatchEncoding
        rows = self._ids(texts, max_length, truncation)
        L = max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (L - len(r)) for r in rows])
        mask = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows])
        return BatchEncoding({"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask})

    def convert_ids_to_tokens(self, ids):
        return [str(i) for i in ids]


def load_retriever(checkpoint: str, device: str):
    """`checkpoint` = HF hub id / local dir (Retriever.from_pretrained, README.md:108), or
    `random:<hidden>:<layers>:<seed>` = random-init VDR towers with the whitespace-id tokenizer
    (there are no weights or WordPiece vocab offline)."""
    from ..ir import Retriever, RetrieverConfig
    if checkpoint.startswith("random:"):
        from ..ir.encoder.vdr import VDREncoder, VDREncoderConfig
        parts = checkpoint.split(":")
        hidden = int(parts[1]) if len(parts) > 1 and parts[1] else 64
        layers = int(parts[2]) if len(parts) > 2 and parts[2] else 2
        torch.manual_seed(int(parts[3]) if len(parts) > 3 else 0)
        kw = dict(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=max(1, hidden // 32), intermediate_size=2 * hidden,
                  vocab_size=30522, max_len=128, topk=768, random_init=True, type="vdr")
        enc_q = VDREncoder(VDREncoderConfig(**kw), tokenizer=WhitespaceIdTokenizer())
        enc_p = VDREncoder(VDREncoderConfig(**{**kw, "max_len": 256}), tokenizer=WhitespaceIdTokenizer())
        model = Retriever(RetrieverConfig(encoder_q=kw, encoder_p=kw), encoder_q=enc_q, encoder_p=enc_p)
    else:
        model = Retriever.from_pretrained(checkpoint)
    return model.to(device).eval()


class Timer:
    def __init__(self):
        self.t0 = time.perf_counter()

    def lap(self) -> float:
        return time.perf_counter() - self.t0
