# bag-of-token chunk walk: pipeline depth S and phase clocks, 21 M docs x 1024 queries
cd $GRAFT_REPO_ROOT
for S in ${SETS:-4 3 5 6}; do
  python3 tools/gen_bq_asm.py $S > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1 || { echo "build failed S=$S"; continue; }
  echo "== S = $S"; VS_PROBE_WALK=6 timeout 300 python3 tools/probe_bot.py 21015324 1024 2>&1 | tail -1
  VS_BP_TIMING=1 VS_PROBE_WALK=6 timeout 300 python3 tools/probe_bot.py 21015324 1024 2>&1 | grep "wave-cycles" | tail -1 | cut -c1-250
done
python3 tools/gen_bq_asm.py 4 > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1
