"""python -m vsearch_amd.inference.search.search_sparse_index --checkpoint=... --query_file=q.jsonl
       --index_file='index*.npz' --save_file=results.json [--batch_size_q=32] [--topk=100] [--device=cuda] [--num_gpus=N]
(examples/inference_sparse/README.md:115-128).  Output: JSON list of {"question", "ids", "scores"}."""
import argparse
import json
import logging

from ..common import Timer, load_retriever, logger, read_jsonl


def run(args, index_type, rerank=False):
    logging.basicConfig(level=logging.INFO)
    queries = read_jsonl(args.query_file)
    model = load_retriever(args.checkpoint, args.device)
    n_gpus = int(getattr(args, "num_gpus", 0) or 0)
    model.load_index(index_file=args.index_file, data_file=getattr(args, "text_file", None), index_type=index_type,
                     devices=n_gpus if n_gpus > 1 else None)
    t = Timer()
    out = []
    for s in range(0, len(queries), args.batch_size_q):
        batch = queries[s:s + args.batch_size_q]
        res = model.retrieve(batch, k=args.topk, rerank=rerank, batch_size=args.batch_size_q)
        for qtext, ids, scores in zip(batch, res.ids.tolist(), res.scores.tolist()):
            out.append({"question": qtext, "ids": ids, "scores": scores})
    dt = t.lap()
    logger.info("***** Searched %d queries in %.2f s (%.1f q/s) *****", len(queries), dt, len(queries) / max(dt, 1e-9))
    with open(args.save_file, "w", encoding="utf-8") as fh:
        json.dump(out, fh)
    return out


def parser(doc):
    ap = argparse.ArgumentParser(description=doc)
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--query_file", required=True)
    ap.add_argument("--index_file", required=True)
    ap.add_argument("--text_file", default=None)
    ap.add_argument("--save_file", required=True)
    ap.add_argument("--batch_size_q", type=int, default=32)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--num_gpus", type=int, default=0, help="row-shard the index over this many GPUs of the node (0 / 1: one device); the shard files are dealt to the GPUs in row order")
    return ap


def main(argv=None):
    return run(parser(__doc__).parse_args(argv), "sparse")


if __name__ == "__main__":
    main()
