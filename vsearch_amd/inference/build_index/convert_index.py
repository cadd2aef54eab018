"""python -m vsearch_amd.inference.build_index.convert_index --index_file='index*.npz' --save_file=index.vsx
       [--index_type=sparse|bag_of_token] [--fp32] [--device=cuda]

Joins scipy `.npz` index shards (what `build_index.sparse_index` / the reference write, index.py:181-202) into ONE
native shard file: the device format verbatim, which `SparseIndex(index_file="index.vsx")` / `--index_file=index.vsx`
load without re-parsing, slicing and stacking the shards (index.py:172-176)."""
import argparse
import logging

from ..common import Timer, logger


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--index_file", required=True, help="glob of .npz shards (row order = sorted file names)")
    ap.add_argument("--save_file", required=True, help="output path, must end in .vsx")
    ap.add_argument("--index_type", default="sparse", choices=["sparse", "bag_of_token"])
    ap.add_argument("--fp32", action="store_true", help="keep fp32 values (default: fp16 like SparseIndex(fp16=True))")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    if not args.save_file.endswith(".vsx"):
        raise SystemExit("--save_file must end in .vsx")
    from ...ir.retriever.index import BoTIndex, SparseIndex
    cls = SparseIndex if args.index_type == "sparse" else BoTIndex
    t = Timer()
    index = cls(args.index_file, None, fp16=not args.fp32, device=args.device)
    index.save(args.save_file)
    n_rows, n_cols = index._shape
    logger.info("***** %s: %d x %d written to %s in %.1f s *****", cls.__name__, n_rows, n_cols, args.save_file, t.lap())
    return args.save_file


if __name__ == "__main__":
    main()
