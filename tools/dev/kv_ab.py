#!/usr/bin/env python3
"""A/B of paged vs contiguous talker KV on one MI355X (Q3_KV_CONTIGUOUS=1 = one extent per row): ms/frame of the bench
session (1.7B, B rows, 512-token prompts), the 4105-position VoiceDesign prefill, and the wall time of a continuous-batching
swap; checks that both layouts give the same codes.   usage: kv_ab.py [--batch 8] [--frames 640] [--reps 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth, api
from common import synthetic_prompt

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="1.7b"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--frames", type=int, default=640)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--no-long", action="store_true")
a = ap.parse_args()
cfg = {"1.7b": q.qwen3_tts_1_7b, "0.6b": q.qwen3_tts_0_6b, "tiny": q.tiny}[a.model]()
model = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=synth.DEFAULT_SEED)
utts = [q.Utterance(synthetic_prompt(512, i), q.Speaker.Ryan, q.Language.English, seed=42 + i) for i in range(a.batch)]
opts = q.SynthesisOptions(max_length=a.frames, eos_token_id=None, seed=42)
ref = {}
for rnd in range(2):
  for contiguous in (0, 1):
    os.environ["Q3_KV_CONTIGUOUS"] = str(contiguous)
    name = "contiguous" if contiguous else "paged"
    best = 1e9; create = 1e9
    for r in range(a.reps):
        t0 = time.perf_counter(); s = model.session(utts, opts); t1 = time.perf_counter()
        s.prefill()
        t2 = time.perf_counter(); s.generate(a.frames, use_graph=True); dt = time.perf_counter() - t2
        codes = np.stack([s.codes(b) for b in range(a.batch)]); s.close()
        best = min(best, dt); create = min(create, t1 - t0)
    same = "first" if "gen" not in ref else ("identical" if (codes == ref["gen"]).all() else "DIFFER")
    ref.setdefault("gen", codes)
    print(f"{name:10s} B={a.batch}: {best * 1e3 / a.frames:.4f} ms/frame ({a.frames} frames), session create {create * 1e3:.2f} ms, codes {same}", flush=True)
    if rnd == 0 and not a.no_long:
        # 4105-position VoiceDesign prompt: prefill, then 64 frames at long context
        lu = [q.Utterance(synthetic_prompt(512, 0), language=q.Language.English, instruct_ids=synthetic_prompt(4096, 5000), seed=42)]
        lo = q.SynthesisOptions(max_length=64, eos_token_id=None, seed=42)
        bp = 1e9; bg = 1e9
        for r in range(a.reps):
            s = model.session(lu, lo); t0 = time.perf_counter(); s.prefill(); t1 = time.perf_counter(); s.generate(64); t2 = time.perf_counter()
            lc = s.codes(0); s.close(); bp = min(bp, t1 - t0); bg = min(bg, t2 - t1)
        same = "first" if "long" not in ref else ("identical" if (lc == ref["long"]).all() else "DIFFER")
        ref.setdefault("long", lc)
        print(f"{name:10s} 4105-position prefill {bp * 1e3:.2f} ms, then {bg * 1e3 / 64:.4f} ms/frame, codes {same}", flush=True)
        # a swap: row 3 of a running B-row session replaced by a 512-token CustomVoice request / by the 4k-prompt request
        s = api.Session(model, utts, opts, frame_budget=a.frames, prompt_budget=4200)
        s.prefill(); s.generate(16)
        for what, u in (("short prompt", q.Utterance(synthetic_prompt(512, 77), q.Speaker.Ryan, q.Language.English, seed=99)), ("4k prompt", lu[0])):
            ts = []
            for r in range(a.reps):
                t0 = time.perf_counter(); s.replace(3, u); ts.append(time.perf_counter() - t0); s.generate(4)
            print(f"{name:10s} replace with a {what}: {min(ts) * 1e3:.3f} ms (best of {a.reps})", flush=True)
        s.close()
print("pool:", model.kv_pool_info())
