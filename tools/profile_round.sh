#!/bin/bash
# One call on the GPU box: the round's evidence set at HEAD -> gpurun_out/<tag>_* (summaries are then copied to profiles/ by
# tools/rocprof_summary.py / pmc_walk_summary.py on the build box).  Usage: bash tools/profile_round.sh r03
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (with secondary configs and CPU baseline)
python3 $ROOT/bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
# 2. kernel stats + HBM traffic of the headline command (separate --pmc passes; no secondary legs under the profiler)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o $TAG -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch -o $TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write -o $TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $OUT/${TAG}_write.log 2>&1
# 2c. HBM traffic of the secondary scan legs (C3, C5, Zipf, fp16 store): FETCH_SIZE / WRITE_SIZE passes of ONE leg each (tools/leg_pmc.py)
for LEG in C3 C5 zipf fp16; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch_$LEG -o ${TAG}$LEG -- python3 $ROOT/tools/leg_pmc.py $LEG > $OUT/${TAG}_fetch_$LEG.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write_$LEG -o ${TAG}$LEG -- python3 $ROOT/tools/leg_pmc.py $LEG > $OUT/${TAG}_write_$LEG.log 2>&1
done
# 2b. the same with the secondary legs (C2 dense, C3, C5, Zipf, embed, head, rerank): kernel rows of the non-search kernels
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats_all -o ${TAG}all -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_prof_all.log 2>&1
# 3. HBM traffic of a 32-query batch on the same index (DESIGN 4's small-batch claim): FETCH_SIZE of the walk launch
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch_b32 -- python3 $ROOT/tools/probe_filter.py 21015324 32 100 fp32 filter > $OUT/${TAG}_fetch_b32.log 2>&1
# 4. utilisation counters of the walk (4 M docs) and its phase clocks
cd $ROOT && bash tools/pmc_walk.sh 4000000 ${TAG}_pmc sq1,sq2,sq3,tcp1,tcc1 > /dev/null 2>&1
# (the summaries: tools/pmc_walk_summary.py gpurun_out/${TAG}_pmc ${TAG} bp_quad_topk)
VS_BP_TIMING=1 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter > $OUT/${TAG}_phase_clocks.txt 2>&1
python3 tools/probe_latency.py 21015324 > $OUT/${TAG}_latency.txt 2>&1
# 5. utilisation of the round's new kernels: the head pre-pass product and the list walk behind it (Zipf, 4 M docs), the bag-of-token chunk walk
VS_PMC_PROBE=probe_zipf.py VS_PMC_ARGS=" " bash tools/pmc_walk.sh 4000000 ${TAG}_pmc_zipf sq1,tcp1,tcc1,tcc2,mfma > /dev/null 2>&1
VS_PMC_PROBE=probe_bot.py VS_PMC_ARGS=" " bash tools/pmc_walk.sh 21015324 ${TAG}_pmc_bot sq1,sq2,tcp1,tcc1 > /dev/null 2>&1
VS_BP_TIMING=1 python3 tools/probe_zipf.py 21015324 1024 > $OUT/${TAG}_zipf_phase_clocks.txt 2>&1
VS_BP_TIMING=1 python3 tools/probe_bot.py 21015324 1024 > $OUT/${TAG}_bot_phase_clocks.txt 2>&1
# 6. summaries on the box (the rocpd databases stay here: gpurun_out/ carries 64 MiB back)
S=$OUT/${TAG}_summaries; mkdir -p $S; cp $ROOT/profiles/pmc_summary.json $S/ 2>/dev/null
db() { find $OUT/$1 -name "*_results.db" | head -1; }
python3 tools/rocprof_summary.py --stats "$(db ${TAG}_stats)" --fetch "$(db ${TAG}_fetch)" --write "$(db ${TAG}_write)" --tag $TAG --queries 1024 --out $S \
    --note "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary (stats); --steps 1 --warmup 0 (pmc passes)" > /dev/null
python3 tools/rocprof_summary.py --stats "$(db ${TAG}_stats_all)" --tag ${TAG}_secondary --out $S --note "bench.py --steps 3 --warmup 1 --no-cpu-baseline (all legs)" > /dev/null
for LEG in C3:C3_1m_sparse C5:C5_bot_21m zipf:zipf_21m fp16:fp16_21m; do
  python3 tools/rocprof_summary.py --fetch "$(db ${TAG}_fetch_${LEG%%:*})" --write "$(db ${TAG}_write_${LEG%%:*})" --tag ${TAG}_${LEG%%:*} --leg ${LEG##*:} --queries 1024 --searches 2 --out $S \
      --note "tools/leg_pmc.py ${LEG%%:*} under rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)" > /dev/null
done
export VS_PMC_SUMMARY_JSON=$S/pmc_summary.json
python3 tools/pmc_walk_summary.py $OUT/${TAG}_pmc ${TAG} bp_quad_topk > /dev/null; python3 tools/pmc_walk_summary.py $OUT/${TAG}_pmc_zipf ${TAG}_head_gemm head_gemm "4 M docs zipf, 1024 queries" > /dev/null
python3 tools/pmc_walk_summary.py $OUT/${TAG}_pmc_zipf ${TAG}_zipf_walk bp_walk_topk "4 M docs zipf, 1024 queries" > /dev/null; python3 tools/pmc_walk_summary.py $OUT/${TAG}_pmc_bot ${TAG}_bq bp_bq_topk "21 M docs bag-of-token, 1024 queries" > /dev/null
cp $ROOT/profiles/${TAG}_*utilisation.txt $S/ 2>/dev/null
find $OUT -name "*_results.db" -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
(cd $ROOT && timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; tail -3 $OUT/${TAG}_pytest_gpu.log)
find $OUT -name "${TAG}*_results.db" | head -20
