# the quad walk's epilogue variants (VS_QUAD_EPI 0 / 1 / 2), 21 M docs x 1024 queries, with phase clocks
cd $GRAFT_REPO_ROOT
for E in ${EPIS:-0 1 2}; do
  touch vsearch_amd/csrc/bp_search.hip; make -C vsearch_amd/csrc -j16 EXTRA=-DVS_QUAD_EPI=$E > /dev/null 2>&1 || { echo "build failed E=$E"; continue; }
  echo "== VS_QUAD_EPI = $E"; VS_PROBE_REPS=4 timeout 300 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter,csr 2>&1 | grep -v "^csr" | tail -2 | cut -c1-200
  VS_BP_TIMING=1 VS_PROBE_REPS=1 timeout 300 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter 2>&1 | grep "wave-cycles" | tail -1 | cut -c1-260
done
touch vsearch_amd/csrc/bp_search.hip; make -C vsearch_amd/csrc -j16 > /dev/null 2>&1
