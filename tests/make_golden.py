"""Generates tests/golden/*.json|npz from the CPU oracle (no GPU). Committed fixtures = regression pins
of the oracle and reference points for the GPU path. Re-run: python tests/make_golden.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import qwen3_tts_rs_amd as q
import oracle as O
from common import oracle_model, synthetic_prompt

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
os.makedirs(OUT, exist_ok=True)

# 1. PCG stream KAT: first 16 rand_f32 for seeds {0, 42, 12345} (sampling.rs:32-51, 84-94)
pcg = {}
for seed in (0, 42, 12345):
    st = ctypes.c_uint64(); O.olib.q3o_rng_seed(seed, ctypes.byref(st))
    pcg[str(seed)] = [float(O.olib.q3o_rng_next(ctypes.byref(st))) for _ in range(16)]
json.dump(pcg, open(os.path.join(OUT, "pcg_kat.json"), "w"), indent=1)

# 2. sampler KATs on the benches/sampling.rs:12-17 logits pattern 5*sin(0.1*i), vocab 3072
V = 3072
base = (5.0 * np.sin(0.1 * np.arange(V))).astype(np.float32)
kats = []
for (T, k, p, rep) in [(0.9, 50, 0.9, 1.05), (0.7, 50, 0.9, 1.0), (1.0, 0, 0.9, 1.05), (0.9, 10, 1.0, 1.5), (0.0, 50, 0.9, 1.05), (1.2, 200, 0.95, 1.1)]:
    for seed in (1, 2, 3, 4):
        rng = np.random.default_rng(seed)
        lg = (base + 0.5 * rng.standard_normal(V)).astype(np.float32)
        seen = (rng.random(V) < 0.03).astype(np.uint8)
        work = lg.copy()
        O.olib.q3o_apply_penalties(O.ptr(work), V, O.ptr(seen), rep, 3, 2, 2150)
        st = ctypes.c_uint64(); O.olib.q3o_rng_seed(100 + seed, ctypes.byref(st))
        tok = O.olib.q3o_sample(O.ptr(work), V, T, k, p, ctypes.byref(st))
        kats.append({"temperature": T, "top_k": k, "top_p": p, "repetition_penalty": rep, "noise_seed": seed, "rng_seed": 100 + seed, "token": int(tok)})
json.dump(kats, open(os.path.join(OUT, "sampler_kat.json"), "w"), indent=1)

# 3. tiny-config end-to-end: codes + PCM for two sampling configs (synthetic checkpoint seed 1234)
cfg = q.tiny()
om = oracle_model(cfg, seed=1234)
e2e = {}
arrays = {}
for name, opts in (("greedy", q.SynthesisOptions(max_length=12, temperature=0.0, seed=42, eos_token_id=None)),
                   ("default", q.SynthesisOptions(max_length=12, seed=42, eos_token_id=None))):
    utt = q.Utterance(synthetic_prompt(20, 0), q.Speaker.Ryan, q.Language.English)
    s = O.OracleSession(om, utt, opts)
    hid, lg = s.prefill_out()
    codes = s.generate()
    pcm = om.decode(codes)
    arrays[f"{name}_codes"] = codes.astype(np.uint32)
    arrays[f"{name}_pcm"] = pcm.astype(np.float32)
    arrays[f"{name}_prefill_logits"] = lg.astype(np.float32)
    e2e[name] = {"frames": int(len(codes)), "pcm_samples": int(len(pcm))}
    s.close()
np.savez_compressed(os.path.join(OUT, "tiny_e2e.npz"), **arrays)
json.dump({"config": "tiny", "checkpoint_seed": 1234, "prompt": "synthetic_prompt(20, 0)", "runs": e2e}, open(os.path.join(OUT, "tiny_e2e.json"), "w"), indent=1)
print("golden written to", OUT, {k: v.shape for k, v in arrays.items()})

# 4./5. full-shape vectors (SURVEY.md §8c fixtures 4 and 5): Qwen3-TTS-1.7B synthetic checkpoint (seed = synth.DEFAULT_SEED),
# 8-frame free run (ids + the smallest top-2 logit margin of the run), the prefill logits, and a full-size vocoder decode
# of those frames; full-size speaker-encoder embedding of a fixed clip. Takes a few minutes of CPU.
if "--full" in sys.argv:
    import time
    from qwen3_tts_rs_amd import synth
    t0 = time.time()
    cfg = q.qwen3_tts_1_7b()
    om = oracle_model(cfg, seed=synth.DEFAULT_SEED)
    print(f"1.7B oracle loaded in {time.time() - t0:.0f}s")
    full = {}
    meta = {"config": "qwen3_tts_1_7b", "checkpoint_seed": synth.DEFAULT_SEED, "prompt": "synthetic_prompt(32, 0)", "frames": 8, "runs": {}}
    for name, opts in (("greedy", q.SynthesisOptions(max_length=8, temperature=0.0, seed=42, eos_token_id=None)),
                       ("default", q.SynthesisOptions(max_length=8, seed=42, eos_token_id=None))):
        utt = q.Utterance(synthetic_prompt(32, 0), q.Speaker.Ryan, q.Language.English)
        s = O.OracleSession(om, utt, opts)
        hid, lg = s.prefill_out()
        codes, tl, cl = s.generate(capture=True)
        srt = np.sort(tl.astype(np.float64), axis=1); tm = float((srt[:, -1] - srt[:, -2]).min())
        srt = np.sort(cl.astype(np.float64), axis=2); cm = float((srt[..., -1] - srt[..., -2]).min())
        full[f"{name}_codes"] = codes.astype(np.uint32)
        full[f"{name}_prefill_logits"] = lg.astype(np.float32)
        full[f"{name}_prefill_hidden"] = hid.astype(np.float32)
        meta["runs"][name] = {"min_talker_top2_margin": tm, "min_cp_top2_margin": cm}
        s.close()
    full["greedy_pcm"] = om.decode(full["greedy_codes"]).astype(np.float32)
    om.close()
    scfg = q.SpeakerEncoderConfig(enc_dim=2048)
    from qwen3_tts_rs_amd.speaker import SpeakerEncoder, synthetic_speaker_checkpoint
    enc = SpeakerEncoder(scfg, device=-1)
    osp = O.OracleSpeakerEncoder(scfg)
    for nme, arr in synthetic_speaker_checkpoint(enc, synth.DEFAULT_SEED):
        osp.set_tensor(nme, arr)
    enc.close()
    tt = np.arange(24000) / 24000.0
    clip = (0.3 * np.sin(2 * np.pi * 220.0 * tt) + 0.2 * np.sin(2 * np.pi * 1330.0 * tt + 1.0)) * (0.5 + 0.5 * np.sin(2 * np.pi * 3.0 * tt))
    full["spk_clip"] = clip.astype(np.float32)
    full["spk_embedding"] = osp.encode(full["spk_clip"])
    full["spk_mel_first_frames"] = O.mel_speaker(full["spk_clip"])[:, :4].copy()
    osp.close()
    np.savez_compressed(os.path.join(OUT, "full_1_7b.npz"), **full)
    json.dump(meta, open(os.path.join(OUT, "full_1_7b.json"), "w"), indent=1)
    print(f"full-shape golden written ({time.time() - t0:.0f}s)", {k: v.shape for k, v in full.items()}, meta["runs"])
