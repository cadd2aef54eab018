#!/bin/bash
# what bank conflicts cost the list walk.  Three builds of the same instruction stream -- default, bank-aware list
# order (postings_arrange = 1), conflict-free addresses (throw-away build #4: -DVS_BP_EXPERIMENT, VS_BP_KNOB=16; 32 = two lanes per bank) --
# each timed at 21 M docs and counted (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS_ATOMIC) at 4 M docs.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/conflicts
mkdir -p $OUT
cd $ROOT
./tools/microbench/bin/lds_conflicts > $OUT/lds_conflicts.txt 2>&1
for cfg in default arrange knob16 knob32; do
  unset VS_BP_KNOB VS_PROBE_ARRANGE
  case $cfg in
    arrange) export VS_PROBE_ARRANGE=1;;
    knob16) export VS_BP_KNOB=16;;
    knob32) export VS_BP_KNOB=32;;
  esac
  python3 tools/probe_filter.py ${DOCS_T:-21015324} 1024 100 fp32 filter > $OUT/time_$cfg.txt 2>&1
  bash tools/pmc_walk.sh 4000000 conflicts/pmc_$cfg sq2 > /dev/null 2>&1
done
tail -n 3 $OUT/time_*.txt
cat $OUT/lds_conflicts.txt
