#!/bin/bash
# round 3, job 6u: 192-channel k = 7 conv as one 3-wave workgroup of 192 co x 64 t (x staged once) vs two of 96 co x 128 t
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" "Q3_CONV_192=1"; do echo "== $v"; env $v python tools/prof_decode.py 640 5 | tail -1; done
Q3_CONV_192=1 bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
grep "launch order" -A70 gpurun_out/vocprof/vocoder_T640.txt | grep " 10[4-9] \| 11[0-3] " | cut -c1-130
