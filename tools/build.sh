#!/bin/bash
# build libvsearch_hip.so; with an argument: also print the register / spill figures of the kernels matching it
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/vsearch_amd/csrc > /tmp/build.log 2>&1 || { grep -E "error|Error|undefined" -A3 /tmp/build.log | head -30; echo BUILD FAILED; exit 1; }
if [ -n "$1" ]; then
  mkdir -p /tmp/dis
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include --cuda-device-only -S $R/vsearch_amd/csrc/bp_search.hip -o /tmp/dis/ci.s 2>&1 | grep error || true
  grep -A45 "\.name:\s*.*$1" /tmp/dis/ci.s | grep -E "\.name|group_segment_fixed|spill|vgpr_count|private_segment_fixed" || true
fi
ls -la $R/vsearch_amd/libvsearch_hip.so
