#!/bin/bash
# round 6 / B2: the frame captured under the prefill + hipGraph instantiated lazily + streaming read-ahead on the own queue:
# streaming / batcher / variant tests, TTFA breakdown; PMC traffic of the GEMV shapes with the write-through transport.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -x -q -k "stream or chunk or batcher or variants or free_run or continuous or ragged or abi or c_host or examples or stages_a_4k" > gpurun_out/r6/b2_tests.txt 2>&1
tail -5 gpurun_out/r6/b2_tests.txt
python tools/dev/eos_mix_ab.py 2 > gpurun_out/r6/b2_eos_mix.txt 2>&1; Q3_BAT_NO_STAGE=1 python tools/dev/eos_mix_ab.py 2 >> gpurun_out/r6/b2_eos_mix.txt 2>&1; cat gpurun_out/r6/b2_eos_mix.txt
python tools/dev/ttfa_breakdown.py 7 > gpurun_out/r6/b2_ttfa.txt 2>&1; tail -2 gpurun_out/r6/b2_ttfa.txt
Q3_AQL=0 python tools/dev/ttfa_breakdown.py 7 > gpurun_out/r6/b2_ttfa_aql0.txt 2>&1; tail -2 gpurun_out/r6/b2_ttfa_aql0.txt
bash tools/pmc_collect.sh 8 > gpurun_out/r6/b2_pmc_collect.log 2>&1; cp gpurun_out/pmc/pmc_gemv_M8.json gpurun_out/r6/r6_pmc_gemv_M8.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6/r6_pmc_gemv_M8.json"))["shapes"]
for k,v in d.items(): print(k, v["algorithmic_bytes"], round(v["fetch_bytes_corrected"]), round(v["write_bytes"]), round((v["fetch_bytes_corrected"]+v["write_bytes"])/v["algorithmic_bytes"],3))
PY
