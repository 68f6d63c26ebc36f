#!/bin/bash
# round 3, job 7a: full GPU suite + smoke on the final state
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
