#!/usr/bin/env python3
"""k_sample alone, 60 launches on 8 rows of bench-like logits (development aid, run under rocprofv3 --kernel-trace): the kernel's
duration outside the frame. Q3_SAMPLE_SLOW_TOPK selects the variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import qwen3_tts_rs_amd as q
V, rows = 3072, 8
rng = np.random.default_rng(5)
opts = q.SynthesisOptions(seed=1)
logits = (3.5 * rng.standard_normal((rows, V))).astype(np.float32)
seen = (rng.random((rows, V)) < 0.05).astype(np.uint8)
us = rng.random(rows).astype(np.float32)
for i in range(60):
    got = q.sample(logits, us, opts, seen=seen, token_count=5)
print(got)
