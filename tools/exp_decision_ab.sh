# walk kernel ms at 21 M docs x 1024 queries for library variants / lock-step windows (one box, back to back)
cd $GRAFT_REPO_ROOT
cp vsearch_amd/libvsearch_hip.so /tmp/lib_head.so
run() { VS_PROBE_REPS=4 python tools/probe_filter.py 21015324 1024 100 fp32 filter 2>&1 | grep "^filter\|workgroup time"; }
for v in ${VARIANTS:-head Hprev head}; do
  if [ $v = head ]; then cp /tmp/lib_head.so vsearch_amd/libvsearch_hip.so; else cp tools/microbench/bin/variants/lib$v.so vsearch_amd/libvsearch_hip.so; fi
  for pace in ${PACES:--1}; do echo "== $v pace=$pace"; VS_BP_PACE=$pace run; done
done
cp /tmp/lib_head.so vsearch_amd/libvsearch_hip.so
