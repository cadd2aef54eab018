# the LDS-tiled head product (VS_HEAD_SHAPE=222) next to the direct one, and its ablations (WRONG results): tools/microbench/bin/lib_head_*.so
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-4000000}
VS_PROBE_REPS=3 timeout 300 python3 tools/probe_zipf.py $N 1024 2>&1 | grep "^zipf" | sed "s/^/direct: /" | cut -c1-170
VS_HEAD_SHAPE=222 VS_PROBE_REPS=3 timeout 300 python3 tools/probe_zipf.py $N 1024 2>&1 | grep "^zipf" | sed "s/^/lds ring: /" | cut -c1-170
cp vsearch_amd/libvsearch_hip.so /tmp/orig.so
for f in tools/microbench/bin/lib_head_*.so; do
  [ -f "$f" ] || continue
  cp $f vsearch_amd/libvsearch_hip.so
  VS_HEAD_SHAPE=222 VS_PROBE_REPS=3 timeout 300 python3 tools/probe_zipf.py $N 1024 2>&1 | grep "^zipf" | sed "s|^|$f: |" | cut -c1-200
done
cp /tmp/orig.so vsearch_amd/libvsearch_hip.so
