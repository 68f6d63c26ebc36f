"""Which prefill path choice changes bits? (development aid) Logits of a 2111-position prompt under env toggles."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import qwen3_tts_rs_amd as q
    from qwen3_tts_rs_amd import synth
    from common import synthetic_prompt
    B = int(sys.argv[2]); n_ins = int(sys.argv[3])
    gm = q.Qwen3TTS.from_synthetic(q.qwen3_tts_0_6b(), seed=synth.DEFAULT_SEED)
    opts = q.SynthesisOptions(max_length=6, seed=3, eos_token_id=None)
    utts = [q.Utterance(synthetic_prompt(12, i), language=q.Language.German, instruct_ids=synthetic_prompt(n_ins, 50 + i), seed=20 + i) for i in range(B)]
    s = gm.session(utts, opts); s.prefill()
    lg = s.get(2, (gm.config.codec_vocab,), b=0)
    np.save(sys.argv[4], lg)
    sys.exit(0)
import numpy as np
n_ins = int(sys.argv[1]) if len(sys.argv) > 1 else 2102
runs = {"b1": (1, {}), "b2": (2, {}), "b1_geo2": (1, {"Q3_GEMM_GEO": "2"}), "b2_geo2": (2, {"Q3_GEMM_GEO": "2"}),
        "b1_geo3": (1, {"Q3_GEMM_GEO": "3"}), "b2_geo3": (2, {"Q3_GEMM_GEO": "3"}),
        "b1_nox3": (1, {"Q3_PREFILL_ATTN_X3": "0"}), "b2_nox3": (2, {"Q3_PREFILL_ATTN_X3": "0"}),
        "b1_nosplit": (1, {"Q3_PREFILL_ATTN_NOSPLIT": "1"}), "b2_nosplit": (2, {"Q3_PREFILL_ATTN_NOSPLIT": "1"}),
        "b2_rows8448": (2, {"Q3_PREFILL_ROWS": "8448"})}
out = {}
for name, (B, env) in runs.items():
    f = f"/tmp/diag_{name}.npy"
    subprocess.run([sys.executable, __file__, "child", str(B), str(n_ins), f], env={**os.environ, **env}, check=True, stderr=subprocess.DEVNULL)
    out[name] = np.load(f)
ref = out["b1"]
for name, v in out.items():
    print(f"{name:14s} max|diff vs b1| = {np.abs(v - ref).max():.3e}")
