# the regression test of the cut decision against the library and against the round-5 code with the same test delay (variant Eknob)
cd $GRAFT_REPO_ROOT
cp vsearch_amd/libvsearch_hip.so /tmp/lib_head.so
echo "== HEAD"; python -m pytest tests/test_gpu_facade.py -q -x -k "late_wave" 2>&1 | tail -3
echo "== Eknob (66f8569: per-thread decision, same delay)"; cp tools/microbench/bin/variants/libEknob.so vsearch_amd/libvsearch_hip.so; python -m pytest tests/test_gpu_facade.py -q -x -k "late_wave" 2>&1 | grep -v "^$" | tail -6
cp /tmp/lib_head.so vsearch_amd/libvsearch_hip.so
VARIANTS="head Hprev" bash tools/exp_decision_ab.sh
