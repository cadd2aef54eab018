cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quad
VS_BP_TIMING=1 python3 tools/probe_filter.py 4000000 1024 100 fp32 filter > gpurun_out/quad/p4m.txt 2>&1; tail -4 gpurun_out/quad/p4m.txt
