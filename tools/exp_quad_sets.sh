# in-situ depth of the quad walk's load pipeline: register sets S (S - 1 chunk loads of a wave in flight), 21 M docs x 1024 queries
cd $GRAFT_REPO_ROOT
for S in ${SETS:-4 5 6 7}; do
  python3 tools/gen_quad_asm.py $S > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1 || { echo "build failed S=$S"; continue; }
  echo "== S = $S"; VS_PROBE_REPS=4 timeout 300 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter 2>&1 | tail -1 | cut -c1-170
done
python3 tools/gen_quad_asm.py 4 > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1
