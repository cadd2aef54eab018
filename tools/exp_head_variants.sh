# what bounds the head pre-pass product: variants of the generated item (tools/gen_head_asm.py VARIANT; WRONG results), 4 M docs zipf x 1024 queries
cd $GRAFT_REPO_ROOT
for V in ${VARIANTS:-"" nostore noa nob nomfma noa_nob noa_nob_nostore}; do
  python3 tools/gen_head_asm.py vsearch_amd/csrc/bp_head_asm.h "$V" > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1 || { echo "build failed $V"; continue; }
  echo "== variant '$V'"; VS_PROBE_REPS=3 timeout 300 python3 tools/probe_zipf.py ${DOCS:-4000000} 1024 2>&1 | tail -1
done
python3 tools/gen_head_asm.py > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1
