"""python -m vsearch_amd.inference.score.eval_wiki21m --result_file=results.json --text_file=corpus.jsonl --qa_file=nq-test.qa.csv
(examples/inference_sparse/README.md:134-141).  Top-k retrieval accuracy: a question counts as hit@k when one
of its first k passages contains an answer string (DrQA-style token match, src/ir/utils/qa_utils.py:258-291)."""
import argparse
import ast
import csv
import json
import unicodedata

import regex

from ..common import read_jsonl

_TOKEN = regex.compile(r"([\p{L}\p{N}\p{M}]+)|([^\p{Z}\p{C}])", flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE)


def _normalize(text: str) -> str:
    return unicodedata.normalize("NFD", text.replace("’", "'").replace("\n", " "))


def _words(text: str):
    return [m.group().lower() for m in _TOKEN.finditer(_normalize(text))]


def has_answer(answers, text, match_type: str = "string") -> bool:
    """qa_utils.py:258-291: uncased token-sequence containment ("string") or regex search ("regex")."""
    if match_type == "regex":
        t = _normalize(text)
        for a in answers:
            try:
                if regex.compile(_normalize(a), flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE).search(t):
                    return True
            except Exception:
                continue
        return False
    words = _words(text)
    for a in answers:
        aw = _words(a)
        for i in range(0, len(words) - len(aw) + 1):
            if aw == words[i:i + len(aw)]:
                return True
    return False


def parse_qa_csv_file(path):
    with open(path, encoding="utf-8") as fh:
        return [(row[0], ast.literal_eval(row[1])) for row in csv.reader(fh, delimiter="\t")]


def evaluate(results, texts, qa, ks=(1, 5, 10, 20, 100)):
    hits = {k: 0 for k in ks}
    for res, (_, answers) in zip(results, qa):
        first = None
        for rank, doc_id in enumerate(res["ids"]):
            passage = texts[doc_id]
            body = passage["text"] if isinstance(passage, dict) else passage
            if has_answer(answers, body):
                first = rank
                break
        for k in ks:
            hits[k] += first is not None and first < k
    n = max(1, len(results))
    return {f"top{k}": 100.0 * hits[k] / n for k in ks}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--result_file", required=True)
    ap.add_argument("--text_file", required=True)
    ap.add_argument("--qa_file", required=True)
    args = ap.parse_args(argv)
    acc = evaluate(json.load(open(args.result_file)), read_jsonl(args.text_file), parse_qa_csv_file(args.qa_file))
    print(json.dumps(acc))
    return acc


if __name__ == "__main__":
    main()
