"""`python -m inference.score.eval_wiki21m` -> vsearch_amd.inference.score.eval_wiki21m (same arguments)."""
from vsearch_amd.inference.score.eval_wiki21m import *  # noqa: F401,F403
from vsearch_amd.inference.score.eval_wiki21m import main

if __name__ == "__main__":
    main()
