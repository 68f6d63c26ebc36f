#!/bin/bash
# usage: small_vs_tiled.sh T full|tiny
mkdir -p /tmp/svt
python tools/dev/small_vs_tiled.py /tmp/svt/a.npz $1 $2 
Q3_CONV_NO_SMALL=1 python tools/dev/small_vs_tiled.py /tmp/svt/b.npz $1 $2 
python - <<'PY'
import numpy as np
a=np.load('/tmp/svt/a.npz'); b=np.load('/tmp/svt/b.npz')
for k in a.files:
    d=np.abs(a[k]-b[k]); print(k, a[k].shape, 'max diff', d.max(), 'n diff', int((d>0).sum()))
PY
