"""python -m vsearch_amd.inference.search.beta_search --checkpoint=... --query_file=q.jsonl --text_file=corpus.jsonl
       --index_file=bow.npz --save_file=results.json [--rerank]      (test/svdr_wiki21m/beta_search.sh)
SVDR "beta search": parametric query x binary bag-of-token index, optional rerank of the k hits."""
from .search_sparse_index import parser, run


def main(argv=None):
    ap = parser(__doc__)
    ap.add_argument("--rerank", action="store_true")
    args = ap.parse_args(argv)
    return run(args, "bag_of_token", rerank=args.rerank)


if __name__ == "__main__":
    main()
