#!/bin/bash
# round 3, job 6p: matrix-core utilisation of the vocoder kernels after the fused unit / geometry changes; C host with the batcher
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
sed -i 's/name = d\[1\].replace("void q3::", "")/name = d[1].replace("(anonymous namespace)::", "").replace("void q3::", "")/' tools/pmc_vocoder.sh
bash tools/pmc_vocoder.sh 640 | cut -c1-150
cp gpurun_out/pmc/vocoder_mfma_T640.txt gpurun_out/r6p_pmc_vocoder_mfma_T640.txt
timeout 600 python -m pytest tests/test_c_host.py -m gpu -x -q 2>&1 | tail -2
