# what one rank of an 8-GPU step costs besides its walk: 2 626 916-doc shard, 1024 queries, every kernel of a search
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VS_PROBE_REPS=5 python3 $R/tools/probe_filter.py 2626916 1024 100 fp32 filter 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/shard_step -o shard -- python3 $R/tools/probe_filter.py 2626916 1024 100 fp32 filter > /dev/null 2>&1
python3 - <<'PY'
import sqlite3, glob, os
db = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/shard_step/**/*.db"), recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, (end - start), start from kernels order by start").fetchall()
# the last search: from the last bp_count_colfreq launch on
last = max(i for i, r in enumerate(rows) if "colfreq" in r[0])
t0 = rows[last][2]
tot = 0
for name, dur, st in rows[last:]:
    print(f"{(st - t0) / 1e3:10.1f} us  {dur / 1e3:10.1f} us  {name[:90]}")
    tot += dur
print(f"kernel time {tot / 1e3:.1f} us; span {(rows[-1][2] + rows[-1][1] - t0) / 1e3:.1f} us")
PY
