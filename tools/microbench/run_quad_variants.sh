cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/row; O=gpurun_out/row/quad.txt; : > $O
for v in "" _s4 _s7 _cxx _nolds _noload _nolds_noload; do for m in "x" "l1" "x rnd"; do echo "== quad_walk$v $m" >> $O; timeout 60 ./tools/microbench/bin/quad_walk$v 24 6630 $m 2>&1 | grep "walk:" | tail -1 >> $O; done; done
echo "== table copy only" >> $O; ./tools/microbench/bin/quad_walk 24 6630 2>&1 | grep "copy only" | tail -1 >> $O
cat $O
