#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box: kernel stats + HBM traffic (separate --pmc passes) -> gpurun_out/<tag>_{stats,fetch,write}
# Usage: bash tools/profile_bench.sh <tag> [extra bench.py args]
TAG=${1:-r02}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o $TAG -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/${TAG}_bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch -o $TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write -o $TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/${TAG}_write.log 2>&1
find $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write -name "*.db" | head
