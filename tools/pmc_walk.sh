#!/bin/bash
# PMC passes over the postings walk (tools/probe_walk.py, 21 M docs, 1024 queries): one rocprofv3 run per counter group
# (VS_PMC_PROBE=probe_bot.py VS_PMC_ARGS=" " : the bag-of-token walk instead) (SQ: 8 counters per pass; TCC: 4), outputs under gpurun_out/pmc_walk/<group>/.  Usage on the GPU box: bash tools/pmc_walk.sh [docs] [tag] [groups: all | sq1,sq2,...]
DOCS=${1:-21015324}
TAG=${2:-pmc_walk}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/${VS_PMC_PROBE:-probe_filter.py} $DOCS ${VS_PMC_B:-1024} ${VS_PMC_ARGS:-100 fp32 filter} > $OUT/$name.log 2>&1
}
ONLY=${3:-all}
want() { [ "$ONLY" = all ] || echo ",$ONLY," | grep -q ",$1,"; }
run_if() { if want $1; then run "$@"; fi; }
run_if sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run_if sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC
run_if sq3 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_SCA
run_if tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run_if tcp2 TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
run_if tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run_if tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_sum
run_if ta1 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run_if mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run_if grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_if fetch FETCH_SIZE
ls -R $OUT | head -50
