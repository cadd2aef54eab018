"""python -m vsearch_amd.inference.build_index.binary_token_index --text_file=corpus.jsonl --save_file=bow.npz
       [--batch_size=32] [--num_shift=999]      (test/svdr_wiki21m/build_binary_token_index.sh:1-3)
Tokenizer-only build of the SVDR bag-of-token index (no encoder forward)."""
from .sparse_index import main as _main


def main(argv=None):
    return _main(argv, index_type="bag_of_token")


if __name__ == "__main__":
    main()
