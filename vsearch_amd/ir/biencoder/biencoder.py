"""Two-tower wrapper with the reference's interface (/root/reference/src/ir/biencoder/biencoder.py:15-123)."""
from __future__ import annotations

import logging
from typing import Dict, List, Union

from transformers import PretrainedConfig, PreTrainedModel

from ..encoder.types import CONFIG_TYPES, ENCODER_TYPES
from ..encoder.vdr import VDREncoderConfig

logger = logging.getLogger(__name__)


class BiEncoderConfig(PretrainedConfig):
    """encoder_q / encoder_p: config dicts with a ``type`` key selecting the tower class (biencoder.py:15-41)."""

    def __init__(self, encoder_q: Dict[str, any] = None, encoder_p: Dict[str, any] = None, max_len=512,
                 shared_encoder=False, device=None, **kwargs):
        self.encoder_q = encoder_q
        self.encoder_p = encoder_p
        self.max_length = max_len
        self.shared_encoder = shared_encoder
        self.device = device
        super().__init__(**kwargs)


class BiEncoder(PreTrainedModel):
    config_class = BiEncoderConfig

    def __init__(self, config: BiEncoderConfig, encoder_q=None, encoder_p=None, **kwargs):
        super().__init__(config)
        self.config = config
        self.encoder_q = encoder_q if encoder_q is not None else self._make_tower(config.encoder_q)
        if config.shared_encoder:
            self.encoder_p = self.encoder_q
            if encoder_p is None and config.encoder_p is not None:
                self.encoder_q.config.max_len = max(self.encoder_q.config.max_len, config.encoder_p.get("max_len", VDREncoderConfig().max_len))
        else:
            self.encoder_p = encoder_p if encoder_p is not None else self._make_tower(config.encoder_p)
        self.default_batch_size = None
        self.post_init()                                   # transformers >= 5: from_pretrained relies on the bookkeeping done here

    @staticmethod
    def _make_tower(cfg: Dict[str, any]):
        kind = cfg["type"]
        if kind not in ENCODER_TYPES:
            raise NotImplementedError(f"encoder type {kind!r} is outside the vocabulary-space retrieval path (only 'vdr' is built)")
        tower_cfg = CONFIG_TYPES[kind](**cfg)
        return ENCODER_TYPES[kind](tower_cfg)

    def forward(self, q_ids, q_segments, q_attn_mask, p_ids, p_segments, p_attn_mask):
        return self.encoder_q(q_ids, q_segments, q_attn_mask), self.encoder_p(p_ids, p_segments, p_attn_mask)

    def encode_queries(self, queries: List[str], batch_size=None, convert_to_tensor=True, **kwargs):
        """Query embeddings without forcing lexical dims (biencoder.py:75-86)."""
        batch_size = batch_size or self.default_batch_size
        return self.encoder_q.embed(queries, batch_size, convert_to_tensor=convert_to_tensor, activate_lexical=False, **kwargs)

    def encode_corpus(self, corpus: Union[List[str], List[Dict[str, str]]], batch_size=None, max_len=None, to_cpu=False,
                      convert_to_tensor=True, **kwargs):
        """Passage embeddings; dict passages become "title [SEP] text" (biencoder.py:88-109)."""
        batch_size = batch_size or self.default_batch_size
        flat = []
        for p in corpus:
            if isinstance(p, dict):
                flat.append(f"{p['title']} [SEP] {p['text']}" if p.get("title") else p["text"])
            elif isinstance(p, str):
                flat.append(p)
        return self.encoder_p.embed(flat, batch_size, max_len=max_len, to_cpu=to_cpu, convert_to_tensor=convert_to_tensor,
                                    activate_lexical=False, **kwargs)

    def encode_corpus_csr(self, corpus: Union[List[str], List[Dict[str, str]]], batch_size=None, max_len=None):
        """`encode_corpus(...)` batch by batch as CSR (encoder.embed_csr: mask stage + to_sparse_csr fused, no dense masked batch)."""
        batch_size = batch_size or self.default_batch_size
        flat = []
        for p in corpus:
            if isinstance(p, dict):
                flat.append(f"{p['title']} [SEP] {p['text']}" if p.get("title") else p["text"])
            elif isinstance(p, str):
                flat.append(p)
        return self.encoder_p.embed_csr(flat, batch_size, max_len=max_len, activate_lexical=False)

    def explain(self, q, p, topk=768, visual=False, max_words=100, log_scale=True, save_file=None):
        """Per-token contribution q_w * p_w, largest first (biencoder.py:111-123)."""
        if visual:
            raise NotImplementedError("word-cloud rendering is out of scope")
        q_dst, p_dst = self.encoder_q.dst(q, topk=topk), self.encoder_p.dst(p, topk=topk)
        prod = {t: q_dst[t] * p_dst[t] for t in q_dst.keys() & p_dst.keys() if q_dst[t] * p_dst[t] != 0}
        return dict(sorted(prod.items(), key=lambda kv: kv[1], reverse=True))
