#!/bin/bash
# round 6 / E3: the last tree's suite + default bench line (the PMC / trace / vocoder / prefill tables of E1 were taken on the tree
# two commits earlier: no kernel changed since — E1's next-node experiment was removed again, host-side changes only).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final6; mkdir -p $O
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/suite.txt; grep -E "passed|failed" $O/suite.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
head -c 400 $O/bench.json; echo
