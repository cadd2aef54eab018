# what building the postings copy of a skewed (Zipf) 21 M-doc index costs, kernel by kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/zipf_build -o z -- python3 $R/tools/probe_zipf.py ${1:-21015324} 64 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3, glob, os, collections
db = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/zipf_build/**/*.db"), recursive=True)[0]
con = sqlite3.connect(db)
d = collections.defaultdict(list)
for n, t in con.execute("select name, (end - start) from kernels"): d[n].append(t / 1e6)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{sum(v):9.1f} ms  {len(v):4d} x {sum(v) / len(v):8.2f} ms  {n[:100]}")
PY
