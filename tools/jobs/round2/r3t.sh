#!/bin/bash
# r3t: refreshed vocoder tables: kernel trace of one 640-frame decode + MFMA-busy PMC
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/prof_decode.py 640 5 2>&1 | tail -1 > gpurun_out/r3t.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3t_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r3t_prof 4 2>&1 | grep -v "gemv\|gather_rows\|copyBuffer\|pack_conv\|fillBuffer" > gpurun_out/r3t_kernels.txt; rm -rf gpurun_out/r3t_prof
cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3t_pmc -o p -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/pmc_mfma_table.py gpurun_out/r3t_pmc 2 "vocoder decode, T = 640 frames, 2 decodes" 2>&1 | grep -v "gemv\|gather_rows\|copyBuffer\|pack_conv" > gpurun_out/r3t_pmc_table.txt; find gpurun_out/r3t_pmc -name "*.csv" -size +5M -delete
cat gpurun_out/r3t.txt gpurun_out/r3t_kernels.txt gpurun_out/r3t_pmc_table.txt
