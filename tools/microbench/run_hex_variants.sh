cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/hex; O=gpurun_out/hex/hex_loop.txt; : > $O
B=tools/microbench/bin
echo "== quad_walk (256-byte chunks, 8 slots), 6208 lists, 2 % links" >> $O
hipcc -O3 --offload-arch=gfx950 -I vsearch_amd/csrc tools/microbench/quad_walk.hip -o $B/quad_walk 2>>$O
timeout 60 $B/quad_walk 24 6208 x arr 2 2>&1 | grep "walk:" | tail -1 >> $O
for S in 4 5 6 7 8 10; do
  python3 tools/gen_hex_asm.py $S /tmp/bp_hex_asm.h >/dev/null && mkdir -p /tmp/hx$S && cp /tmp/bp_hex_asm.h /tmp/hx$S/ && cp vsearch_amd/csrc/bp_hex_loop.h /tmp/hx$S/
  hipcc -O3 --offload-arch=gfx950 -I /tmp/hx$S tools/microbench/hex_walk.hip -o $B/hex_walk_$S 2>>$O
  for pct in 3 7; do echo "== hex_walk S=$S, $pct % links" >> $O; timeout 60 $B/hex_walk_$S 24 776 x arr $pct 2>&1 | grep "walk:" | tail -1 >> $O; done
done
for v in nolds noload nolink; do
  python3 tools/gen_hex_asm.py 6 /tmp/bp_hex_asm.h $v >/dev/null && mkdir -p /tmp/hx$v && cp /tmp/bp_hex_asm.h /tmp/hx$v/ && cp vsearch_amd/csrc/bp_hex_loop.h /tmp/hx$v/
  hipcc -O3 --offload-arch=gfx950 -I /tmp/hx$v tools/microbench/hex_walk.hip -o $B/hex_walk_$v 2>>$O
  echo "== hex_walk S=6 variant $v (0 % links)" >> $O; timeout 60 $B/hex_walk_$v 24 776 x arr 0 2>&1 | grep "walk:" | tail -1 >> $O
done
echo "== hex_walk S=6 l1-resident chunks" >> $O; timeout 60 $B/hex_walk_6 24 776 l1 arr 0 2>&1 | grep "walk:" | tail -1 >> $O
echo "== hex_walk S=6 random banks" >> $O; timeout 60 $B/hex_walk_6 24 776 x rnd 3 2>&1 | grep "walk:" | tail -1 >> $O
cat $O
