# The round-5 wrong-result bug, on the library itself: variants of libvsearch_hip.so built from csrc with the bag-of-token walk's next
# block base loaded AHEAD of the walk statement (Eold: the library at 66f8569^, the commit that failed in round 5; E1: that change re-applied to today's bag-of-token walk; E3: to the quad walk; E2: E1 with s_waitcnt vmcnt(0) as the statement's first
# instruction), under tools/contention_check.py with 4 processes.  Variants are built in the container (tools/microbench/bin/variants/).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/vgpr; O=gpurun_out/vgpr/library_variants.txt; : > $O
cp vsearch_amd/libvsearch_hip.so /tmp/lib_head.so
for v in ${VARIANTS:-E1 E2 head}; do
  if [ $v = head ]; then cp /tmp/lib_head.so vsearch_amd/libvsearch_hip.so; else cp tools/microbench/bin/variants/lib$v.so vsearch_amd/libvsearch_hip.so; fi
  for opts in ${OPTS:-postings_packed=0 postings_packed=1}; do
    echo "== variant $v, $opts, ${PROCS:-4} processes x ${REPS:-200} searches" >> $O
    VS_CHECK_OPTS=$opts timeout 600 python tools/contention_check.py ${PROCS:-4} ${REPS:-200} ${KIND:-bot} 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cp /tmp/lib_head.so vsearch_amd/libvsearch_hip.so
cat $O
