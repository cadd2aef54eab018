# times the encoder's mask stage on experimental builds of the library (tools/microbench/bin/lib_mr_*.so, e.g. built with -DMR_TIMING)
# usage: run_mr_variants.sh [B [V]]
cd ${GRAFT_REPO_ROOT:-.}
cp vsearch_amd/libvsearch_hip.so /tmp/orig.so
for f in tools/microbench/bin/lib_mr_*.so; do
  cp $f vsearch_amd/libvsearch_hip.so
  echo "== $f"; timeout 200 python3 tools/probe_mask.py ${1:-1024} ${2:-29523} 2>&1 | grep "mask stage\|GB/s"
done
cp /tmp/orig.so vsearch_amd/libvsearch_hip.so
