#!/usr/bin/env python3
"""Where a streaming session's time to first audio goes (1.7B, one utterance, chunk = 10 frames): session create, prefill, the first
generate call (graph capture + instantiate + 10 replays), a second generate call (10 replays), the first chunk's vocoder.
   usage: ttfa_breakdown.py [reps=5]     (run ON the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b(), device=0, seed=synth.DEFAULT_SEED)
u = [q.Utterance(synth.synthetic_prompt(512, 0), q.Speaker.Ryan, q.Language.English, seed=42)]
rows = []
for r in range(reps + 1):
    opts = q.SynthesisOptions(max_length=30, eos_token_id=None, seed=42, chunk_frames=10)
    t0 = time.perf_counter(); s = m.session(u, opts)
    t1 = time.perf_counter(); s.prefill()
    t2 = time.perf_counter(); s.generate(10, use_graph=True); n, _ = s.frames(0)
    t3 = time.perf_counter(); s.generate(10, use_graph=True); n, _ = s.frames(0)
    t4 = time.perf_counter(); codes = s.codes(0)[:10]; m.decode_codes(codes)
    t5 = time.perf_counter(); s.close()
    if r: rows.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))])
    ta = time.perf_counter(); ss = m.synthesize_streaming(u[0].text_ids, q.Speaker.Ryan, q.Language.English, opts); ss.next_chunk()
    tb = time.perf_counter(); ss._s.close()
    if r: rows[-1].append((tb - ta) * 1e3)
med = np.median(np.array(rows), axis=0)
print("median ms over %d reps: create %.2f | prefill %.2f | first generate(10) %.2f | second generate(10) %.2f | decode 10 frames %.2f | streaming first chunk (TTFA) %.2f"
      % (reps, *med))
print("capture + instantiate ~ %.2f ms" % (med[2] - med[3]))
