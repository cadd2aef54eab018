cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/row; O=gpurun_out/row/quad_links.txt; : > $O
for pct in 0 7 30; do echo "== quad_walk, $pct % of the chunks linked" >> $O; timeout 60 ./tools/microbench/bin/quad_walk 24 6208 x arr $pct 2>&1 | grep "walk:" | tail -1 >> $O; done
cat $O
