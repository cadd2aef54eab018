"""VDR text encoder with the reference's interface (/root/reference/src/ir/encoder/vdr.py:21-192).

BERT body and the vocabulary projection GEMM run in PyTorch-ROCm (hipBLASLt / MFMA); everything after
the GEMM -- elu1p∘max-pool (vdr.py:73-75), top-k mask, bag-of-word mask, OR, multiply
(vdr.py:152-169) -- runs in the HIP kernels of libvsearch_hip.so.  Inference only: the HIP head has
no backward, so ``require_grad=True`` raises.
"""
from __future__ import annotations

import logging
from typing import List, Union

import torch
import torch.nn.functional as F
from transformers import AutoModel, AutoTokenizer, BertConfig, BertModel, PreTrainedModel

from ..utils import sparse as sp

logger = logging.getLogger(__name__)


class VDREncoderConfig(BertConfig):
    """BERT config + the VDR knobs (vdr.py:21-44; defaults as conf/biencoder/vdr.yaml:5-23)."""

    def __init__(self, model_id="bert-base-uncased", max_len=256, norm=False, shift_vocab_num=999,
                 topk=768, pooling="max", pooling_topk=None, random_init=False, **kwargs):
        super().__init__(**kwargs)
        self.model_id = model_id
        self.max_len = max_len
        self.norm = norm
        self.shift_vocab_num = shift_vocab_num
        self.topk = topk
        self.pooling = pooling
        self.pooling_topk = pooling_topk
        self.random_init = random_init      # build the BERT body from this config instead of downloading weights


class VDREncoder(PreTrainedModel):
    config_class = VDREncoderConfig

    def __init__(self, config: VDREncoderConfig, bert_model=None, tokenizer=None, **kwargs):
        super().__init__(config, **kwargs)
        self.config = config
        self.ln = torch.nn.LayerNorm(config.hidden_size)
        if bert_model is not None:
            self.bert_model = bert_model
        elif getattr(config, "random_init", False):
            self.bert_model = BertModel(config, add_pooling_layer=False)
        else:
            self.bert_model = AutoModel.from_pretrained(config.model_id, add_pooling_layer=False)
        self.tokenizer = tokenizer if tokenizer is not None else (
            None if getattr(config, "random_init", False) else AutoTokenizer.from_pretrained(config.model_id))
        self.post_init()

    # -- helpers -------------------------------------------------------------------------------------
    def build_bow_mask(self, input_ids):
        return sp.build_bow_mask(input_ids, vocab_size=self.config.vocab_size, shift_num=self.config.shift_vocab_num,
                                 norm=self.config.norm)

    def _vocab_logits(self, input_ids, token_type_ids, attention_mask):
        out = self.bert_model(input_ids=input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask)
        hidden = self.ln(out.last_hidden_state)
        w = self.bert_model.embeddings.word_embeddings.weight[self.config.shift_vocab_num:, :]
        return hidden @ w.t()                                      # [B, L, V]: MFMA GEMM territory

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        """[B, L] token ids -> [B, V] lexical representation (vdr.py:58-84). Pad positions are pooled
        like the reference (no attention mask in the max)."""
        if self.config.pooling == "mean" and not self.config.pooling_topk:
            # vdr.py:80 reads an undefined name: the reference cannot run this configuration either
            raise NotImplementedError('pooling="mean" needs pooling_topk (the reference\'s plain-mean branch is dead code, vdr.py:80)')
        if self.config.pooling not in ("max", "mean"):
            raise NotImplementedError
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise NotImplementedError("vsearch_amd implements the inference path only (HIP head has no backward)")
        with torch.no_grad():
            out = self.bert_model(input_ids=input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask)
            hidden = self.ln(out.last_hidden_state)
            if not hidden.is_cuda:
                raise RuntimeError("VDREncoder runs on an MI355X: move the encoder with .to('cuda') (no CPU fallback)")
            w = self.bert_model.embeddings.word_embeddings.weight[self.config.shift_vocab_num:, :]
            if self.config.pooling == "mean":
                # vdr.py:76-79: mean of the pooling_topk largest activations per vocabulary dimension
                emb = sp.head_pool_mean_topk(hidden @ w.t(), int(self.config.pooling_topk))
                return F.normalize(emb) if self.config.norm else emb
            if hidden.shape[-1] % 32 == 0 and hidden.dtype == torch.float32 and hidden.shape[1] > 64:
                # passage-shaped batches: fused projection + max-pool + elu1p, no [B, L, V] logits (1.9 GB at 64 x 256);
                # short query batches are faster through the library GEMM + vs_head_pool
                emb = sp.head_project_pool(hidden, w)
            else:
                emb = sp.head_pool(hidden @ w.t())
            return F.normalize(emb) if self.config.norm else emb

    def encode(self, texts: Union[List[str], str], max_len: int = None):
        max_len = max_len or self.config.max_len
        texts = [texts] if isinstance(texts, str) else texts
        enc = self.tokenizer.batch_encode_plus(texts, padding=True, truncation=True, max_length=max_len, return_tensors="pt")
        return enc.to(self.device)

    def embed(self, texts: Union[List[str], str], batch_size: int = 128, max_len: int = None, topk: int = None,
              bow: bool = False, activate_lexical: bool = True, require_grad: bool = False, to_cpu: bool = False,
              convert_to_tensor: bool = True, show_progress_bar: bool = False, **kwargs):
        """Texts -> [N, V] lexical representations (vdr.py:97-179).

        topk == 0: only the dims of present tokens; topk in (None -> config.topk, -1): all dims / top-k;
        bow: binary token vector; activate_lexical: force the present tokens' dims on.
        """
        if require_grad:
            raise NotImplementedError("vsearch_amd implements the inference path only (require_grad=True is training)")
        max_len = max_len or self.config.max_len
        topk = topk if topk is not None else self.config.topk
        texts = [texts] if isinstance(texts, str) else texts
        was_training = self.training
        if was_training:
            self.eval()
        V = self.config.vocab_size - self.config.shift_vocab_num
        chunks = []
        starts = range(0, len(texts), batch_size)
        if show_progress_bar:
            from tqdm import tqdm
            starts = tqdm(starts)
        with torch.no_grad():
            for s in starts:
                enc = self.encode(texts[s:s + batch_size], max_len=max_len)
                ids = enc["input_ids"]
                if bow:
                    emb = torch.empty((ids.shape[0], V), dtype=torch.float32, device=ids.device)
                    if not emb.is_cuda:
                        emb = emb.cuda()
                    sp.apply_embed_mask_(emb, ids, self.config.vocab_size, self.config.shift_vocab_num, 0, True, bow=True)
                    if self.config.norm:
                        emb = F.normalize(emb)
                else:
                    emb = self(**enc).contiguous()
                    sp.apply_embed_mask_(emb, ids if activate_lexical else None, self.config.vocab_size,
                                         self.config.shift_vocab_num, topk, activate_lexical)
                chunks.append(emb)
        out = torch.cat(chunks, dim=0)
        if not convert_to_tensor:
            out = out.cpu().numpy()
        elif to_cpu:
            out = out.cpu()
        if was_training:
            self.train()
        return out

    def embed_csr(self, texts: Union[List[str], str], batch_size: int = 128, max_len: int = None, topk: int = None,
                  activate_lexical: bool = True):
        """Generator over the batches of `embed(texts, ...)`, each as CSR: (rowptr int64 [b+1], cols int32, vals fp32, V) CUDA tensors --
        what `embed(batch).to_sparse_csr()` holds (vdr.py:97-179 + retriever.py:304), with the mask stage and the CSR conversion fused
        in one kernel (`vs_embed_mask_to_csr`): the masked dense batch is never written.  Falls back to embed + dense_to_csr outside
        the fused kernel's range (topk <= 0, V > 32 Ki, topk + L > 8192, a non-fp32 head; `norm` needs no fallback: forward() has normalised already).
        One divergence from the reference, for NON-FINITE activations only: `batch_emb *= mask` (vdr.py:169) turns an unselected NaN / inf
        into NaN, which to_sparse_csr() keeps as a stored element; the fused kernel emits the SELECTED non-zero cells only, so such a cell
        is dropped here (the unfused path, x * 0, keeps it).  Finite activations: identical (tests/test_gpu_facade.py)."""
        max_len = max_len or self.config.max_len
        topk = topk if topk is not None else self.config.topk
        texts = [texts] if isinstance(texts, str) else texts
        was_training = self.training
        if was_training:
            self.eval()
        V = self.config.vocab_size - self.config.shift_vocab_num
        try:
            with torch.no_grad():
                for s in range(0, len(texts), batch_size):
                    enc = self.encode(texts[s:s + batch_size], max_len=max_len)
                    emb = self(**enc).contiguous()
                    fused = (topk is not None and int(topk) > 0 and V <= 32768 and emb.dtype == torch.float32
                             and min(V, int(topk) + (enc["input_ids"].shape[1] if activate_lexical else 0)) <= sp.FUSED_CSR_MAX_KEPT)
                    if fused:
                        rp, ci, va = sp.embed_mask_to_csr(emb, enc["input_ids"], self.config.vocab_size, self.config.shift_vocab_num, int(topk), activate_lexical)
                    else:
                        sp.apply_embed_mask_(emb, enc["input_ids"] if activate_lexical else None, self.config.vocab_size, self.config.shift_vocab_num,
                                             topk, activate_lexical)
                        rp, ci, va = sp.dense_to_csr(emb)
                    yield rp, ci, va, V
        finally:
            if was_training:
                self.train()

    def disentangle(self, text: str, topk: int = 768, visual=False, save_file=None):
        """Top-k (token, weight) pairs of a text's representation (vdr.py:181-192)."""
        if visual:
            raise NotImplementedError("word-cloud rendering is out of scope")
        top = self.embed(text).topk(topk)
        ids = [i + self.config.shift_vocab_num for i in top.indices.flatten().tolist()]
        return dict(zip(self.tokenizer.convert_ids_to_tokens(ids), top.values.flatten().tolist()))

    dst = disentangle
