# the fused mask -> CSR and the mask stage: hipEvent kernel time next to rocprofv3's
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/tools/probe_mask_csr.py 1024 29523 768 1 2>&1 | grep "equal\|ms"
python3 $R/tools/probe_mask.py 1024 29523 2>&1 | grep "equal\|GB/s"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mask_prof -o m -- python3 $R/tools/probe_mask_csr.py 1024 29523 768 1 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3, glob, os
db = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/mask_prof/**/*.db"), recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, (end - start) from kernels").fetchall()
import collections
d = collections.defaultdict(list)
for n, t in rows: d[n].append(t)
for n, v in d.items():
    if "mask_rows" in n: v.sort(); print(f"{n[:80]}: n={len(v)} median {v[len(v)//2] / 1e3:.1f} us min {v[0] / 1e3:.1f} us")
PY
