# the cut-decision regression test, repeated (it is a timing test: one pass proves little)
cd $GRAFT_REPO_ROOT
for i in $(seq ${REPS:-6}); do python -m pytest tests/test_gpu_facade.py -q -x -k "late_wave" 2>&1 | tail -1; done
