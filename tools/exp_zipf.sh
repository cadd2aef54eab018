# Zipf 21 M-doc leg: head pre-pass shapes against the in-walk strips
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 600 python3 tools/leg_pmc.py zipf 2 2>&1 | tail -1 | python3 -c "
import json,sys
r=list(json.loads(sys.stdin.read()).values())[0]
print({k:r[k] for k in ('ms_per_step','queries_per_sec','scan_kernel_ms','head_gemm_ms','scan_launches_per_search','fallback_queries','head_columns','postings_copy_bytes','index_build_s')})"; }
for cfg in ${CFGS:-"VS_HEAD_SHAPE=24" "VS_HEAD_SHAPE=18" "VS_HEAD_SHAPE=42" "VS_HEAD_SHAPE=14" "VS_BP_HEAD_GEMM=0"}; do run $cfg; done
