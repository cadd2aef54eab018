# N concurrent processes of tools/microbench/vgpr_across_asm per mode: gpurun_out/vgpr/vgpr_across_asm.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/vgpr tools/microbench/bin; O=gpurun_out/vgpr/vgpr_across_asm.txt; : > $O
hipcc -O3 --offload-arch=gfx950 -I vsearch_amd/csrc tools/microbench/vgpr_across_asm.hip -o tools/microbench/bin/vgpr_across_asm 2>>$O || exit 1
for mode in ${MODES:-0 1 2 3}; do
  for procs in ${PROCS:-1 2 4 8}; do
    echo "== mode $mode, $procs processes" >> $O
    pids=""
    for p in $(seq $procs); do timeout 120 tools/microbench/bin/vgpr_across_asm $mode ${ITERS:-4000} ${LAUNCHES:-4} ${STREAMS:-1} >> $O.$p 2>&1 & pids="$pids $!"; done
    for p in $pids; do wait $p; done
    for p in $(seq $procs); do cat $O.$p >> $O; rm -f $O.$p; done
  done
done
cat $O
