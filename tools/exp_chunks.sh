# the quad walk's block chunks: items per CU against the sweep's locality (21 M docs; B = 256 .. 2048), and run-to-run stability in fresh processes
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-21015324}
for i in 1 2 3 4; do VS_PROBE_REPS=4 python3 tools/probe_filter.py $N 1024 100 fp32 filter 2 2>&1 | grep "^filter" | sed "s/^/B=1024 process $i: /"; done
VS_PROBE_REPS=4 python3 tools/probe_filter.py $N 2048 100 fp32 filter 1,2,4 2>&1 | grep "^filter" | sed "s/^/B=2048: /"
VS_PROBE_REPS=4 python3 tools/probe_filter.py $N 512 100 fp32 filter 2,4,8 2>&1 | grep "^filter" | sed "s/^/B=512: /"
VS_PROBE_REPS=4 python3 tools/probe_filter.py $N 256 100 fp32 filter 4,8,16 2>&1 | grep "^filter" | sed "s/^/B=256: /"
VS_PROBE_REPS=4 python3 tools/probe_filter.py $N 128 100 fp32 filter 8,16,32 2>&1 | grep "^filter" | sed "s/^/B=128: /"
VS_PROBE_REPS=4 python3 tools/probe_filter.py $N 1024 100 fp16 filter 2,4 2>&1 | grep "^filter" | sed "s/^/fp16 B=1024: /"
