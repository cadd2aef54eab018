cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/row; O=gpurun_out/row/align.txt; : > $O
for v in "" _nolds_novalu; do for al in 1 2 4 8 16 32; do for m in x l1; do echo "== row_walk$v $m align $((al*8)) B" >> $O; ./tools/microbench/bin/row_walk$v 24 6208 $m $al 2>&1 | grep "mode 0" | tail -1 >> $O; done; done; done; cat $O
