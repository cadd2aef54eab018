cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quad
for walk in -1 0; do
  echo "== postings_walk $walk"
  VS_PROBE_WALK=$walk VS_PROBE_REPS=250 timeout 200 python3 tools/probe_filter.py 21015324 1024 100 fp32 filter > gpurun_out/quad/pw_$walk.txt 2>&1 &
  PID=$!
  sleep 16
  for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature .*junction|hotspot|edge" | sed 's/=*//g' | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 3; done
  wait $PID; tail -1 gpurun_out/quad/pw_$walk.txt | cut -c1-160
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
