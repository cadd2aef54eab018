# bag-of-token chunk walk: chunk shape (lanes x dwords per list), pipeline depth S, block rows; phase clocks; 21 M docs x 1024 queries
#   SETS="4x4:7:0 8x2:4:0 8x1:4:2048" bash tools/exp_bq_sets.sh        (shape:S:postings_rows, 0 = auto)
cd $GRAFT_REPO_ROOT
for CFG in ${SETS:-4x4:4:0 4x4:7:0 8x2:4:0 8x2:7:0}; do
  IFS=: read SHAPE S ROWS <<< "$CFG"
  python3 tools/gen_bq_asm.py $S vsearch_amd/csrc/bp_bq_asm.h $SHAPE > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1 || { echo "build failed $CFG"; continue; }
  echo "== shape $SHAPE S = $S rows $ROWS"; VS_PROBE_WALK=6 timeout 300 python3 tools/probe_bot.py 21015324 1024 $ROWS 2>&1 | tail -1
  VS_BP_TIMING=1 VS_PROBE_WALK=6 timeout 300 python3 tools/probe_bot.py 21015324 1024 $ROWS 2>&1 | grep "wave-cycles" | tail -1 | cut -c1-250
done
python3 tools/gen_bq_asm.py > /dev/null && make -C vsearch_amd/csrc -j16 > /dev/null 2>&1
