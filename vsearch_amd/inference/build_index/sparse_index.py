"""python -m vsearch_amd.inference.build_index.sparse_index --checkpoint=... --text_file=corpus.jsonl
       --save_file=index.npz [--batch_size=64] [--num_shard=N --shard_id=i] [--device=cuda]
(flags of examples/inference_sparse/README.md:71-107; output: scipy CSR .npz, one per shard)"""
import argparse
import logging

from ..common import Timer, load_retriever, logger, read_jsonl, shard_slice


def main(argv=None, index_type="sparse"):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--checkpoint", required=index_type != "bag_of_token", default="random:64:2:0")
    ap.add_argument("--text_file", required=True)
    ap.add_argument("--save_file", required=True)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--num_shard", type=int, default=1)
    ap.add_argument("--shard_id", type=int, default=0)
    ap.add_argument("--num_shift", type=int, default=999)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    t_all = Timer()
    texts = read_jsonl(args.text_file)
    texts = texts[shard_slice(len(texts), args.num_shard, args.shard_id)]
    model = load_retriever(args.checkpoint, args.device)
    t = Timer()
    model.build_index(texts, batch_size=args.batch_size, index_type=index_type)
    logger.info("***** Finish Indexing *****")
    logger.info("***** Time for indexing (exclude i/o): %.0f s *****", t.lap())
    model.save_index(args.save_file)
    logger.info("***** Time for indexing (include i/o): %.0f s *****", t_all.lap())
    info = model.index._device_index().info()
    logger.info("***** Index save to: %s *****", args.save_file)
    logger.info("***** Index matrix shape: (%d, %d) *****", info.n_rows, info.n_cols)
    logger.info("***** Index sparsity rate: %.2f%% *****", 100.0 * info.nnz / max(1, info.n_rows * info.n_cols))
    return model.index


if __name__ == "__main__":
    main()
