"""The C ABI driven by a host with no Python and no torch in the process (examples/c_host.c): compiled with gcc against
include/q3tts.h, linked to libq3tts.so only. It loads seeded weights, joins a world-size-1 RCCL communicator through
q3_dp_* (RCCL + HIP runtime resolved from /opt/rocm in that process — the deployment a Rust host would have), runs
one synthesis and dumps codes + WAV; the same model driven from Python must give the same bytes."""
import ctypes
import os
import subprocess
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_compiles_against_the_header(tmp_path):
    exe = tmp_path / "c_host"
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_host.c"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "qwen3_tts_rs_amd"), "-lq3tts",
                           "-Wl,-rpath," + os.path.join(ROOT, "qwen3_tts_rs_amd")])
    assert exe.exists()


@pytest.mark.gpu
def test_c_host_matches_python_host(tmp_path):
    import qwen3_tts_rs_amd as q
    from qwen3_tts_rs_amd import _lib, api, synth
    from qwen3_tts_rs_amd.config import CConfig
    exe = tmp_path / "c_host"
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_host.c"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "qwen3_tts_rs_amd"), "-lq3tts",
                           "-Wl,-rpath," + os.path.join(ROOT, "qwen3_tts_rs_amd")])
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe), str(tmp_path), "6"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frames 6 samples 11520" in r.stdout
    c_codes = api.load_codes_binary(str(tmp_path / "c_host_codes.bin"))
    # the same model from the Python host
    c = CConfig(); _lib.check(_lib.lib.q3_config_default(0, ctypes.byref(c)))
    c.text_vocab, c.text_dim, c.hidden, c.inter, c.n_layers = 512, 128, 128, 256, 2
    c.n_heads, c.n_kv_heads, c.cp_hidden, c.cp_inter, c.cp_layers, c.cp_heads, c.cp_kv_heads = 2, 1, 128, 256, 2, 2, 1
    cfg = q.Q3Config.from_c(c)
    m = q.Qwen3TTS(cfg, 0)
    for name, n, stored in synth.manifest(m._h):
        is_norm = ("norm.weight" in name) or ("cluster_usage" in name)
        out = np.empty(n, np.uint16 if stored == synth.BF16 else np.float32)
        _lib.check(_lib.lib.q3_synth_fill(7, name.encode(), stored, 0.01 if is_norm else 0.04, 1.0 if is_norm else 0.0, n, out.ctypes.data_as(ctypes.c_void_p)))
        m.set_tensor(name, out, stored)
    m.finalize()
    text = [(17 * i + 3) % 512 for i in range(12)]
    opts = q.SynthesisOptions(max_length=6, seed=42, eos_token_id=None)
    s = m.session([q.Utterance(text, q.Speaker.Ryan, q.Language.English)], opts)
    audio, _ = s.run()
    codes = s.codes(0).copy(); s.close(); m.close()
    assert (codes == c_codes).all()
    with wave.open(str(tmp_path / "c_host.wav")) as w:
        assert (w.getnframes(), w.getframerate(), w.getsampwidth()) == (11520, 24000, 2)
        pcm16 = np.frombuffer(w.readframes(11520), dtype="<i2")
    np.testing.assert_array_equal(pcm16, api.pcm16(audio[0].samples))
