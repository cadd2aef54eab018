"""python -m vsearch_amd.inference.search.search_dense_index ... (examples/inference_dense/README.md): dense .pt index."""
from .search_sparse_index import parser, run


def main(argv=None):
    return run(parser(__doc__).parse_args(argv), "dense")


if __name__ == "__main__":
    main()
