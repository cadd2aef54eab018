cd $GRAFT_REPO_ROOT
for p in 4 8 16 32 64; do echo "== direct, pace $p"; VS_BP_PACE=$p timeout 200 python3 tools/probe_bot.py 21015324 1024 2>&1 | tail -1 | cut -c1-150; done
