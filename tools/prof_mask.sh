cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r04_mask_fetch -- python3 $R/tools/probe_mask.py 1024 > $R/gpurun_out/r04_mask_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r04_mask_write -- python3 $R/tools/probe_mask.py 1024 > $R/gpurun_out/r04_mask_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_mask_stats -- python3 $R/tools/probe_mask.py 1024 > $R/gpurun_out/r04_mask_stats.log 2>&1
ls $R/gpurun_out/r04_mask_fetch/*/ | head
