for pw in 0 1 2 4; do echo pace $pw; VS_BP_PACE=$pw VS_PROBE_WALK=2 timeout 120 python3 tools/probe_filter.py 4000000 1024 100 fp32 filter,csr 2>&1 | grep -v "amdgpu.ids\|^csr" | cut -c1-200; done
