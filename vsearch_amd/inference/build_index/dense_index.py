"""python -m vsearch_amd.inference.build_index.dense_index ... (examples/inference_dense/README.md): dense .pt index."""
from .sparse_index import main as _main


def main(argv=None):
    return _main(argv, index_type="dense")


if __name__ == "__main__":
    main()
