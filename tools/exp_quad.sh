cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quad
cp vsearch_amd/libvsearch_hip.so /tmp/lib_s4.so
for S in 4 5 6 4 5 6; do cp vsearch_amd/libvsearch_hip_s$S.so vsearch_amd/libvsearch_hip.so 2>/dev/null || cp /tmp/lib_s4.so vsearch_amd/libvsearch_hip.so; echo "== S $S rows 1920 pace 8"; VS_BP_PACE=8 VS_PROBE_ROWS=1920 VS_PROBE_REPS=6 timeout 200 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter 2>&1 | tail -1 | cut -c1-200; done
cp /tmp/lib_s4.so vsearch_amd/libvsearch_hip.so
