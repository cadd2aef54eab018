"""`python -m inference.build_index.binary_token_index` -> vsearch_amd.inference.build_index.binary_token_index (same arguments)."""
from vsearch_amd.inference.build_index.binary_token_index import *  # noqa: F401,F403
from vsearch_amd.inference.build_index.binary_token_index import main

if __name__ == "__main__":
    main()
