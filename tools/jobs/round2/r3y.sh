#!/bin/bash
# smoke() on the GPU box, as the driver runs it
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -c "
import __graft_entry__ as g, time
t=time.time(); g.smoke(); print('smoke ok', round(time.time()-t,1),'s')" 2>&1 | tail -3
