cd $GRAFT_REPO_ROOT
for k in 0 4; do echo "== knob $k"; VS_BP_KNOB=$k VS_PROBE_REPS=4 timeout 200 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter 2>&1 | tail -1 | cut -c1-170; done
