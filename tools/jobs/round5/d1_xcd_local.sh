#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5 build/probe
hipcc --offload-arch=gfx950 -O3 tools/hw/xcd_local.hip -o build/probe/xcd_local && timeout 120 build/probe/xcd_local | tee gpurun_out/r5/d1_xcd_local.txt
