"""`python -m inference.search.search_dense_index` -> vsearch_amd.inference.search.search_dense_index (same arguments)."""
from vsearch_amd.inference.search.search_dense_index import *  # noqa: F401,F403
from vsearch_amd.inference.search.search_dense_index import main

if __name__ == "__main__":
    main()
