#!/bin/bash
# the combination table on a five-kernel chain with the code-predictor layer's shapes
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/chain5 build/probe
F="--offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=12"
hipcc $F tools/hw/aql_chain5.hip -o build/probe/aql_chain5 -lhsa-runtime64
hipcc $F --cuda-device-only --no-gpu-bundle-output -c tools/hw/aql_chain5.hip -o build/probe/aql_chain5.hsaco
timeout 240 build/probe/aql_chain5 build/probe/aql_chain5.hsaco | tee gpurun_out/chain5/chain5.txt
