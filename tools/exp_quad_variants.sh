# the quad walk on experimental builds of the library (tools/microbench/bin/lib_quad_*.so) next to the product, 21 M docs
cd ${GRAFT_REPO_ROOT:-.}
run() { VS_PROBE_REPS=4 timeout 400 python3 tools/probe_filter.py ${N:-21015324} 1024 100 fp32 filter 2>&1 | grep "^filter" | cut -c1-150; }
echo "== product"; run; run
cp vsearch_amd/libvsearch_hip.so /tmp/orig.so
for f in tools/microbench/bin/lib_quad_*.so; do
  [ -f "$f" ] || continue
  cp $f vsearch_amd/libvsearch_hip.so
  echo "== $f"; run; run
done
cp /tmp/orig.so vsearch_amd/libvsearch_hip.so
