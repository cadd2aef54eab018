"""Import-path shim: the command lines of the reference's docs (``python -m inference.search.beta_search ...``,
/root/reference/test/svdr_wiki21m/beta_search.sh:5, examples/inference_sparse/README.md:71) run the vsearch_amd CLIs."""
