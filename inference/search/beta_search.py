"""`python -m inference.search.beta_search` -> vsearch_amd.inference.search.beta_search (same arguments)."""
from vsearch_amd.inference.search.beta_search import *  # noqa: F401,F403
from vsearch_amd.inference.search.beta_search import main

if __name__ == "__main__":
    main()
