"""Soak of the native batcher: many random requests (all four prompt kinds, own options) through 8 rows of a tiny model; a sample
is checked against batch-1 runs, free device memory is reported every 100 requests (side sessions come and go: nothing may
accumulate), and the model's KV page pool must be empty again when the batcher is gone (paged KV: pages are taken as rows grow,
relinked on a swap, returned when a row is replaced). Usage: soak_batcher.py [n_requests] [max_text_tokens = 30] [pool_limit_pages = 0]
(max_text_tokens > 128: VoiceDesign instructions of up to that many tokens — those rows span several 128-position pages)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
import qwen3_tts_rs_amd as q
from common import synthetic_prompt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
MAXT = int(sys.argv[2]) if len(sys.argv) > 2 else 30
LIMIT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = q.tiny()
m = q.Qwen3TTS.from_synthetic(cfg)
rng = np.random.default_rng(11)
def request(i):
    kind = int(rng.integers(0, 4)); n_text = int(rng.integers(1, MAXT))
    if kind == 0: u = q.Utterance(synthetic_prompt(n_text, i), q.Speaker.Ryan, q.Language.English)
    elif kind == 1: u = q.Utterance(synthetic_prompt(n_text, i), language=q.Language.German, instruct_ids=synthetic_prompt(int(rng.integers(1, max(20, MAXT))), 50 + i))
    elif kind == 2: u = q.Utterance(synthetic_prompt(n_text, i), language=q.Language.French, xvector=rng.standard_normal(cfg.hidden).astype(np.float32))
    else: u = q.Utterance(synthetic_prompt(n_text, i), language=q.Language.French, xvector=rng.standard_normal(cfg.hidden).astype(np.float32),
                          ref_codes=rng.integers(0, 2048, size=(int(rng.integers(2, 9)), 16)).astype(np.uint32), ref_text_ids=synthetic_prompt(int(rng.integers(1, 6)), 90 + i))
    u.seed = 1000 + i; u.max_length = int(rng.integers(1, 40))
    u.options = q.SynthesisOptions(temperature=float(rng.choice([0.0, 0.7, 1.1])), top_k=int(rng.choice([0, 5, 50])), max_length=40, seed=1,
                                   eos_token_id=None if rng.integers(0, 2) else q.SynthesisOptions().eos_token_id)
    return u
utts = [request(i) for i in range(N)]
if LIMIT: m.kv_pool_limit(LIMIT)
b = q.Batcher(m, slots=8, frame_budget=40, prompt_budget=max(48, MAXT + 24), options=q.SynthesisOptions(max_length=40, seed=1))
t0 = time.time(); tickets = []; results = {}; nxt = 0
while len(results) < N:
    while nxt < N and nxt - len(results) < 24:                        # keep up to 24 requests in the system
        tickets.append(b.submit(utts[nxt], want_pcm=(nxt % 7 == 0))); nxt += 1
    b.step(int(rng.integers(1, 12)))
    for i, t in enumerate(tickets):
        if i not in results and b.poll(t)[0] >= 2:
            results[i] = b.fetch(t)
            if len(results) % 100 == 0:
                free, total = torch.cuda.mem_get_info()
                print(f"{len(results)} done, {time.time() - t0:.1f} s, device memory in use {(total - free) / 2**20:.0f} MiB, kv pool {m.kv_pool_info()}", flush=True)
check = rng.choice(N, size=24, replace=False)
for i in check:
    u = utts[i]
    s1 = m.session([u], u.options); s1.prefill(); s1.generate(100, use_graph=False)
    assert (results[i][0] == s1.codes(0)).all(), i
    if results[i][1] is not None: assert (results[i][1] == s1.decode(0)).all(), i
    s1.close()
print(f"soak OK: {N} requests, {sum(r[0].shape[0] for r in results.values())} frames, {len(check)} checked against batch-1 runs")
b.close()
info = m.kv_pool_info()
assert info["pages_in_use"] == 0, info
print(f"kv pool after close: {info}")
m.close()
