#!/usr/bin/env python3
"""A/B of the frame submission paths on one MI355X: hipGraphLaunch vs the library's own AQL queue (Q3_AQL=1: HIP's packet
headers, Q3_AQL=2: no boundary fences, Q3_AQL=3: fence-free boundaries between the kernels that move their data write-through;
a spec may carry environment settings: `3/Q3_AQL_T_REL=0/Q3_AQL_T_ONLY=sk2`). Same session shape as bench.py (1.7B, B utterances, 512-token prompts); prints
ms/frame per path and checks that every path produces the same codes.   usage: aql_ab.py [--model 1.7b] [--batch 8] [--frames 200]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
from common import synthetic_prompt

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="1.7b"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--modes", default="0,1,2,0,1,2", help="Q3_AQL values; `1:a:r` = mode 1 with Q3_AQL_ACQ=a Q3_AQL_REL=r (fence halves priced separately; results invalid when a fence is off)")
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
cfg = {"1.7b": q.qwen3_tts_1_7b, "0.6b": q.qwen3_tts_0_6b, "tiny": q.tiny}[a.model]()
model = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=synth.DEFAULT_SEED)
utts = [q.Utterance(synthetic_prompt(512, i), q.Speaker.Ryan, q.Language.English, seed=42 + i) for i in range(a.batch)]
opts = q.SynthesisOptions(max_length=a.frames, eos_token_id=None, seed=42)
os.environ["Q3_AQL_VERBOSE"] = "1"
os.environ["Q3_AQL_UNSAFE"] = "1"      # this tool IS the probe: fence-free modes are measured on purpose (their codes are expected to differ)
ref = None
for spec in a.modes.split(","):
    envs = spec.split("/")[1:]
    parts = spec.split("/")[0].split(":"); mode = int(parts[0])
    os.environ["Q3_AQL"] = str(mode)
    for k in ("Q3_AQL_ACQ", "Q3_AQL_REL", "Q3_AQL_T_ACQ", "Q3_AQL_T_REL", "Q3_AQL_T_ONLY", "Q3_AQL_HOST_KERNARG", "Q3_KARG_PREFETCH"): os.environ.pop(k, None)
    for e in envs:
        k, v = e.split("=", 1); os.environ[k] = v
    if len(parts) == 3:
        os.environ["Q3_AQL_ACQ"], os.environ["Q3_AQL_REL"] = parts[1], parts[2]
    best = 1e9
    for r in range(a.reps):
        s = model.session(utts, opts)
        s.prefill()
        t0 = time.perf_counter(); s.generate(a.frames, use_graph=True); dt = time.perf_counter() - t0
        path, nodes = s.submit_info()
        codes = np.stack([s.codes(b) for b in range(a.batch)])
        s.close()
        best = min(best, dt)
    if ref is None: ref = codes
    same = codes.shape == ref.shape and bool((codes == ref).all())
    where = ""
    if not same and codes.shape == ref.shape:
        d = (codes != ref)                                # [B][frames][16]
        first = [int(np.argmax(d[b].any(axis=1))) if d[b].any() else -1 for b in range(a.batch)]
        where = f"  first differing frame per row {first}, {int(d.sum())} of {d.size} codes"
    print(f"Q3_AQL={spec}: path {path} ({nodes} packets/frame)  {best * 1e3 / a.frames:.3f} ms/frame  codes {'identical' if same else 'DIFFER'}{where}", flush=True)
