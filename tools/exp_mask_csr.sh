# the fused mask stage -> CSR: product build, then every experimental build tools/microbench/bin/lib_mr_*.so (e.g. -DMR_TIMING: cycles per phase)
cd ${GRAFT_REPO_ROOT:-.}
echo "== product"; timeout 300 python3 tools/probe_mask_csr.py ${1:-1024} ${2:-29523} 768 1 2>&1 | grep -v Warning
timeout 300 python3 tools/probe_mask.py ${1:-1024} ${2:-29523} 2>&1 | grep "equal\|GB/s\|mask stage"
cp vsearch_amd/libvsearch_hip.so /tmp/orig.so
for f in tools/microbench/bin/lib_mr_*.so; do
  cp $f vsearch_amd/libvsearch_hip.so
  echo "== $f"; timeout 300 python3 tools/probe_mask_csr.py ${1:-1024} ${2:-29523} 768 1 2>&1 | grep "equal\|ms\|cycles\|workgroups"
  timeout 300 python3 tools/probe_mask.py ${1:-1024} ${2:-29523} 2>&1 | grep "equal\|GB/s\|mask stage"
done
cp /tmp/orig.so vsearch_amd/libvsearch_hip.so
