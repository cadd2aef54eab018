"""Inference CLI counterparts (upstream's absent `inference.*` package; flags from
examples/inference_sparse/README.md:71-141 and test/svdr_wiki21m/*.sh)."""
import json

import numpy as np
import pytest

from vsearch_amd.inference.common import shard_slice
from vsearch_amd.inference.score import eval_wiki21m as ev


def test_has_answer_matches_drqa_semantics():
    assert ev.has_answer(["Green Tea"], "Benefits of green  tea: many.")              # uncased, token-wise
    assert not ev.has_answer(["tea leaf"], "green tea leaves")                          # whole tokens only
    assert ev.has_answer(["U.S."], "the U.S. economy")                                  # punctuation tokens
    assert ev.has_answer(["café"], "the café opened")                             # NFD normalisation
    assert ev.has_answer([r"19\d\d"], "born in 1987", match_type="regex")
    assert not ev.has_answer(["x"], "")


def test_eval_topk_accuracy(tmp_path):
    texts = ["alpha beta", {"title": "t", "text": "gamma delta"}, "epsilon", "zeta eta"]
    results = [{"question": "q0", "ids": [2, 1, 0], "scores": [3, 2, 1]}, {"question": "q1", "ids": [0, 2, 3], "scores": [3, 2, 1]}]
    qa = [("q0", ["Delta"]), ("q1", ["omega"])]
    acc = ev.evaluate(results, texts, qa, ks=(1, 2, 3))
    assert acc == {"top1": 0.0, "top2": 50.0, "top3": 50.0}
    (tmp_path / "qa.csv").write_text("q0\t['Delta']\nq1\t['omega']\n")
    assert ev.parse_qa_csv_file(str(tmp_path / "qa.csv")) == qa


def test_shard_slice_partitions():
    for n, s in [(10, 3), (7, 8), (21015324, 8)]:
        parts = [shard_slice(n, s, i) for i in range(s)]
        assert sum(p.stop - p.start for p in parts) == n and parts[0].start == 0 and parts[-1].stop == n


@pytest.mark.gpu
def test_build_search_eval_end_to_end(tmp_path):
    """build_index.sparse_index (2 shards) -> search.search_sparse_index -> score.eval_wiki21m, and the
    bag-of-token build + beta_search with rerank, on a random-init retriever with the id tokenizer."""
    from vsearch_amd.inference.build_index import binary_token_index, sparse_index
    from vsearch_amd.inference.search import beta_search, search_sparse_index
    rng = np.random.default_rng(0)
    docs = [" ".join(map(str, [101] + rng.integers(1996, 6000, size=int(rng.integers(8, 30))).tolist() + [102])) for _ in range(40)]
    (tmp_path / "corpus.jsonl").write_text("".join(json.dumps(d) + "\n" for d in docs))
    queries = [docs[3], docs[17], docs[29]]                       # a passage used as query must retrieve itself
    (tmp_path / "q.jsonl").write_text("".join(json.dumps(q) + "\n" for q in queries))
    ck = "random:64:2:0"
    for sid in range(2):
        sparse_index.main([f"--checkpoint={ck}", f"--text_file={tmp_path/'corpus.jsonl'}", f"--save_file={tmp_path}/index{sid}.npz",
                           "--batch_size=16", "--num_shard=2", f"--shard_id={sid}"])
    res = search_sparse_index.main([f"--checkpoint={ck}", f"--query_file={tmp_path/'q.jsonl'}", f"--index_file={tmp_path}/index*.npz",
                                    f"--save_file={tmp_path/'res.json'}", "--topk=5"])
    assert [r["ids"][0] for r in res] == [3, 17, 29]
    assert json.load(open(tmp_path / "res.json"))[0]["question"] == queries[0]
    from vsearch_amd.inference.build_index import convert_index               # npz shards -> one native shard file -> same results
    convert_index.main([f"--index_file={tmp_path}/index*.npz", f"--save_file={tmp_path/'all.vsx'}", "--fp32"])
    res_v = search_sparse_index.main([f"--checkpoint={ck}", f"--query_file={tmp_path/'q.jsonl'}", f"--index_file={tmp_path/'all.vsx'}",
                                      f"--save_file={tmp_path/'res_v.json'}", "--topk=5"])
    assert [r["ids"] for r in res_v] == [r["ids"] for r in res]
    (tmp_path / "qa.csv").write_text("".join(f"{q}\t{[docs[i].split()[3]]!r}\n" for q, i in zip(queries, (3, 17, 29))))
    from vsearch_amd.inference.score import eval_wiki21m
    acc = eval_wiki21m.main([f"--result_file={tmp_path/'res.json'}", f"--text_file={tmp_path/'corpus.jsonl'}", f"--qa_file={tmp_path/'qa.csv'}"])
    assert acc["top1"] == 100.0
    binary_token_index.main([f"--text_file={tmp_path/'corpus.jsonl'}", f"--save_file={tmp_path/'bow.npz'}", "--batch_size=32"])
    for extra in ([], ["--rerank"]):
        res = beta_search.main([f"--checkpoint={ck}", f"--query_file={tmp_path/'q.jsonl'}", f"--text_file={tmp_path/'corpus.jsonl'}",
                                f"--index_file={tmp_path/'bow.npz'}", f"--save_file={tmp_path/'beta.json'}", "--topk=5"] + extra)
        assert len(res) == 3 and all(len(r["ids"]) == 5 for r in res)
        assert [r["ids"][0] for r in res] == [3, 17, 29]
