"""Pins the C oracle against an independent numpy float64 restatement (tests/np_reference.py) on the
tiny configurations: prefill embeddings/logits, one code-predictor run, one talker step, the vocoder."""
import numpy as np
import pytest

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
import oracle as O
import np_reference as NP
from common import manifest_handle, synthetic_prompt


def _both(cfg, seed):
    h = manifest_handle(cfg)
    om = O.OracleModel(cfg); w = NP.W()
    for name, arr, dt in synth.synthetic_checkpoint(cfg, h, seed):
        om.set_tensor(name, arr, dt); w.add(name, arr, dt)
    om.finalize(3)
    return om, NP.NpModel(cfg, w)


@pytest.fixture(scope="module", params=["tiny", "tiny_same_width"])
def models(request):
    cfg = q.tiny() if request.param == "tiny" else q.tiny_same_width()
    om, npm = _both(cfg, 99)
    yield cfg, om, npm
    om.close()


def test_prefill_and_step(models):
    cfg, om, npm = models
    text = synthetic_prompt(6)
    utt = q.Utterance(text, q.Speaker.Vivian, q.Language.Chinese)
    opts = q.SynthesisOptions(max_length=2, seed=1)
    s = O.OracleSession(om, utt, opts)
    emb = npm.prefill_custom_voice(text, q.Speaker.Vivian.token_id(), q.Language.Chinese.token_id())
    assert emb.shape[0] == s.prefill_len() == 10
    assert np.abs(s.prefill_embeds() - emb).max() < 1e-4 * max(1.0, np.abs(emb).max())
    caches = [{"k": [], "v": []} for _ in range(cfg.n_layers)]
    hid = npm.talker_layers(emb, caches, 0)
    normed, logits = npm.talker_head(hid)
    ohid, olg = s.prefill_out()
    assert np.abs(ohid - normed[-1]).max() < 2e-4
    assert np.abs(olg - logits[0]).max() < 2e-3
    # trailing text (lib.rs:508-519)
    tr, pad = s.trailing()
    ref_tr = np.concatenate([npm.text_proj(text[1:]), npm.text_proj([NP.TTS_EOS])], 0)
    assert np.abs(tr - ref_tr).max() < 1e-4 and np.abs(pad - npm.text_proj([NP.TTS_PAD])[0]).max() < 1e-4
    # one decode step at offset = prefill_len
    rng = np.random.default_rng(0)
    e = rng.standard_normal(cfg.hidden).astype(np.float32)
    oh, ol = s.talker_step(e)
    h2 = npm.talker_layers(e[None].astype(np.float64), caches, 10)
    n2, l2 = npm.talker_head(h2)
    assert np.abs(oh - n2[-1]).max() < 2e-4 and np.abs(ol - l2[0]).max() < 2e-3
    s.close()


def test_code_predictor(models):
    cfg, om, npm = models
    utt = q.Utterance(synthetic_prompt(3))
    s = O.OracleSession(om, utt, q.SynthesisOptions(max_length=1, seed=1))
    rng = np.random.default_rng(1)
    lh = rng.standard_normal(cfg.hidden).astype(np.float32); se = rng.standard_normal(cfg.hidden).astype(np.float32)
    codes, lg = s.cp_generate(lh, se)
    rcodes, rlg = npm.cp_generate(lh.astype(np.float64), se.astype(np.float64))
    # teacher-force check: compare logits group by group while the codes agree
    for g in range(15):
        assert np.abs(lg[g] - rlg[g]).max() < 2e-3, g
        if codes[g] != rcodes[g]:
            srt = np.sort(rlg[g]); assert srt[-1] - srt[-2] < 1e-3
            break
    s.close()


@pytest.mark.parametrize("T", [1, 3])
def test_vocoder(models, T):
    cfg, om, npm = models
    rng = np.random.default_rng(T)
    codes = rng.integers(0, 2048, size=(T, 16)).astype(np.uint32)
    codes[:, 0] = rng.integers(0, 3072, size=T)
    pcm = om.decode(codes)
    ref = npm.decode(codes)
    assert pcm.shape[0] == T * 1920
    assert np.sqrt(np.mean((pcm - ref) ** 2)) < 1e-4


@pytest.mark.parametrize("n_text,n_ref_text,n_ref", [(6, 4, 3), (2, 1, 8), (0, 3, 2)])
def test_icl_prompt(models, n_text, n_ref_text, n_ref):
    """ICL voice-clone prompt (talker.rs:511-564 icl_mode, 646-709 streaming overlay; lib.rs:913-929, 1239-1257)."""
    cfg, om, npm = models
    rng = np.random.default_rng(n_text * 100 + n_ref)
    text = synthetic_prompt(n_text, 3); ref_text = synthetic_prompt(n_ref_text, 4)
    ref_codes = rng.integers(0, 2048, size=(n_ref, 16)).astype(np.uint32); ref_codes[:, 0] = rng.integers(0, 3072, n_ref)
    xv = rng.standard_normal(cfg.hidden).astype(np.float32)
    utt = q.Utterance(text, language=q.Language.Korean, xvector=xv, ref_codes=ref_codes, ref_text_ids=ref_text)
    s = O.OracleSession(om, utt, q.SynthesisOptions(max_length=200, seed=1))
    rp, ml = s.effective()
    assert rp == 1.5 and ml == max(75, 6 * n_text)                      # lib.rs:913-929
    assert s.prefill_len() == 9 + n_ref + 1
    emb = s.prefill_embeds()
    w = npm.w; H = cfg.hidden
    role = npm.text_proj([NP.IM_START, NP.ASSISTANT, NP.NEWLINE])
    pad = npm.text_proj([NP.TTS_PAD]); bos = npm.text_proj([NP.TTS_BOS])
    cod = npm.codec_emb([NP.CODEC_THINK, NP.CODEC_THINK_BOS, q.Language.Korean.token_id(), NP.CODEC_THINK_EOS])
    cod6 = np.concatenate([cod, xv[None].astype(np.float64), npm.codec_emb([NP.CODEC_PAD])], 0)
    ref = np.concatenate([role, np.concatenate([np.repeat(pad, 5, 0), bos], 0) + cod6], 0)
    all_text = npm.text_proj(list(ref_text) + list(text) + [NP.TTS_EOS])
    n_codec = n_ref + 1
    frames = [npm.codec_emb([NP.CODEC_BOS])[0]]
    for f in range(n_ref):
        e = npm.codec_emb([ref_codes[f, 0]])[0]
        for g in range(1, 16):
            e = e + w.g(f"talker.code_predictor.model.codec_embedding.{g - 1}.weight", cfg.cp_vocab, H)[ref_codes[f, g]]
        frames.append(e)
    codec = np.stack(frames)
    if len(all_text) > n_codec:
        icl = all_text[:n_codec] + codec; trailing = all_text[n_codec:]
    else:
        icl = np.concatenate([all_text, np.repeat(pad, n_codec - len(all_text), 0)], 0) + codec; trailing = pad
    ref = np.concatenate([ref, icl], 0)
    assert np.abs(emb - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    tr, _ = s.trailing()
    assert tr.shape[0] == trailing.shape[0] and np.abs(tr - trailing).max() < 2e-4
    s.close()
