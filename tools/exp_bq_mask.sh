# bag-of-token chunk walk: pad cells masked out of the ds_add (v_cmpx_gt_u16_sdwa) or added to spare documents, 21 M docs x 1024 queries
cd $GRAFT_REPO_ROOT
for M in nomask mask; do
  python3 tools/gen_bq_asm.py 4 vsearch_amd/csrc/bp_bq_asm.h $M > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1 || { echo "build failed $M"; continue; }
  echo "== $M"; VS_PROBE_WALK=6 timeout 300 python3 tools/probe_bot.py 21015324 1024 2>&1 | tail -1
  VS_BP_TIMING=1 VS_PROBE_WALK=6 timeout 300 python3 tools/probe_bot.py 21015324 1024 2>&1 | grep "wave-cycles" | tail -1 | cut -c1-250
done
