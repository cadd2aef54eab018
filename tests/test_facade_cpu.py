"""Host-side facade checks that need no GPU: Retriever.save_pretrained / from_pretrained keep the reference's
checkpoint layout (README.md:108 `Retriever.from_pretrained(...)`; module names `encoder_q|p.{ln,bert_model}` as in
src/ir/encoder/vdr.py:50-55, src/ir/biencoder/biencoder.py:50-59)."""
import torch


def test_retriever_pretrained_roundtrip(tmp_path):
    from vsearch_amd.ir import Retriever, RetrieverConfig
    kw = dict(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64, vocab_size=30522, max_len=48, topk=32,
              random_init=True, type="vdr")
    torch.manual_seed(0)
    r = Retriever(RetrieverConfig(encoder_q=kw, encoder_p=kw))
    r.save_pretrained(tmp_path)
    back = Retriever.from_pretrained(tmp_path)
    a, b = r.state_dict(), back.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert {"encoder_q.ln.weight", "encoder_p.ln.weight", "encoder_q.bert_model.embeddings.word_embeddings.weight"} <= set(a)
    assert back.encoder_q.config.topk == 32 and back.config.shared_encoder is False
    shared = Retriever(RetrieverConfig(encoder_q=kw, encoder_p=kw, shared_encoder=True))
    assert shared.encoder_p is shared.encoder_q
