#!/bin/bash
# One call on the GPU box: the round's evidence set at HEAD -> gpurun_out/<tag>_* (summaries are then copied to profiles/ by
# tools/rocprof_summary.py / pmc_walk_summary.py on the build box).  Usage: bash tools/profile_round.sh r03
TAG=${1:-r04}
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
# 2b. the same with the secondary legs (C2 dense, C3, C5, Zipf, embed, head, rerank): kernel rows of the non-search kernels
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats_all -o ${TAG}all -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_prof_all.log 2>&1
# 3. HBM traffic of a 32-query batch on the same index (DESIGN 4's small-batch claim): FETCH_SIZE of the walk launch
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch_b32 -- python3 $ROOT/tools/probe_filter.py 21015324 32 100 fp32 filter > $OUT/${TAG}_fetch_b32.log 2>&1
# 4. utilisation counters of the walk (4 M docs) and its phase clocks
cd $ROOT && bash tools/pmc_walk.sh 4000000 ${TAG}_pmc sq1,sq2,sq3,tcp1,tcc1 > /dev/null 2>&1
# (the summaries: tools/pmc_walk_summary.py gpurun_out/${TAG}_pmc ${TAG} bp_quad_topk)
VS_BP_TIMING=1 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter > $OUT/${TAG}_phase_clocks.txt 2>&1
python3 tools/probe_latency.py 21015324 > $OUT/${TAG}_latency.txt 2>&1
(cd $ROOT && timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; tail -3 $OUT/${TAG}_pytest_gpu.log)
find $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write -name "*.db" | head
