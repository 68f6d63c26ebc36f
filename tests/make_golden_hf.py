"""Third-party cross-check of the CPU oracle (BUILD CONTAINER ONLY — imports `transformers` + `torch`; nothing of this travels
to the GPU box except the small fixture it writes). The oracle (oracle/q3_oracle.c) is the builder's restatement of the
reference's candle-CPU path; the reference itself cannot run here. What CAN run here is Hugging Face's own PyTorch code for
the same published blocks:

  decoder D2-D9   transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe  (Qwen3-Omni Code2Wav: CausalConvNet,
                  CausalTransConvNet, ConvNeXtBlock, SnakeBeta, Code2WavDecoderResidualUnit / DecoderBlock,
                  Code2WavTransformerModel — the 12 Hz codec decoder of Qwen3-TTS is this architecture)
  quantiser D1    transformers.models.mimi.modeling_mimi  (MimiEuclideanCodebook: embed_sum / clamp(cluster_usage))
  talker A4/A2    ...TalkerCodePredictorDecoderLayer (Qwen3 attention: q/k RMSNorm, GQA, rotate-half RoPE, SwiGLU MLP)
  code predictor  ...Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration (round 5): the 15-pass LOOP itself (A3) — which
      A3 / A10    embedding table and which lm_head each pass uses, the two-token first pass, cache positions — driven by the module's
                  own forward() with its own KV cache, greedy (`python tests/make_golden_hf.py cp_loop` -> hf_cp_loop.npz)
  frame loop      ...Qwen3OmniMoeTalkerForConditionalGeneration (round 6): the talker's OUTER loop (A1 / A10, lib.rs:530-656) — upstream's
      A1 / A10    own generate() with its own prepare_inputs_for_generation(): talker step -> code 0 -> its code predictor's generate()
                  -> the next input = sum of the 16 code embeddings + trailing_text_hidden[step] or tts_pad_embed, on its own KV
                  caches (`python tests/make_golden_hf.py frame_loop` -> hf_frame_loop.npz)

The script loads the repo's seeded synthetic checkpoint (tiny config) into those modules, runs the chain stage by stage
and writes tests/golden/hf_crosscheck.npz; tests/test_oracle_vs_hf.py then holds the C oracle to it.

Where the Qwen3-Omni block differs from /root/reference/src/models/codec/*.rs (each handled explicitly below):
  1. CausalTransConvNet trims `kernel - stride` samples from BOTH ends (modeling: left_pad = right_pad = k - s); the
     reference trims on the right only (causal_trans_conv.rs:76-99, "exact input * stride output"). For the decoder blocks
     (k = 2r, s = r) the HF output is therefore the reference's output without its first r samples. The chain below uses
     F.conv_transpose1d + right trim (the reference rule); the HF module's own output is stored too and the test checks the
     shift relation against the oracle.
  2. Code2Wav runs its transformer at the model width with sliding-window attention and no in/out projections; the
     reference's Decoder12Hz projects 1024 -> 512 -> 1024 around it and attends to the full causal prefix
     (decoder_12hz.rs:536-583). The chain adds the two nn.functional.linear projections and sets the window beyond T.
  3. Code2Wav embeds codes with one averaged embedding table; Qwen3-TTS decodes a split residual VQ (decoder_12hz.rs:
     420-452) — the Mimi quantiser code is used for that stage instead.
  4. LayerNorm / GELU / softmax are torch's (two-pass variance, erf GELU): same published definitions, different
     summation orders — hence tolerances, not bit equality.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.nn.functional as F
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
from common import manifest_handle, synthetic_prompt

from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as M
from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import (Qwen3OmniMoeCode2WavConfig,
                                                                              Qwen3OmniMoeTalkerCodePredictorConfig)
from transformers.models.mimi import modeling_mimi as MM
from transformers.models.mimi.configuration_mimi import MimiConfig

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_crosscheck.npz")
SEED, T = 4321, 12
torch.manual_seed(0); torch.set_grad_enabled(False); torch.set_num_threads(4)


def checkpoint(cfg, seed):
    h = manifest_handle(cfg)
    W = {}
    for name, arr, dt in synth.synthetic_checkpoint(cfg, h, seed):
        a = synth.bf16_to_f32(arr) if dt == synth.BF16 else np.asarray(arr, np.float32)
        W[name] = torch.from_numpy(np.array(a, np.float32).reshape(-1))
    q._lib.lib.q3_model_free(h)
    return W


def load(module, W, prefix):
    sd = module.state_dict(); got = 0
    for k, v in sd.items():
        name = prefix + k
        if name in W:
            assert W[name].numel() == v.numel(), (name, W[name].numel(), tuple(v.shape))
            sd[k] = W[name].reshape(v.shape).clone(); got += 1
    module.load_state_dict(sd)
    return got


def main():
    cfg = q.tiny()
    W = checkpoint(cfg, SEED)
    out = {}
    rng = np.random.default_rng(99)
    codes = rng.integers(0, cfg.dec_cb_size, size=(T, 16)).astype(np.int64); codes[:, 0] = rng.integers(0, 3072, T) % cfg.dec_cb_size
    out["codes"] = codes.astype(np.uint32)

    # ---- D1: split residual VQ decode with Mimi's codebook arithmetic ----
    mcfg = MimiConfig(codebook_size=cfg.dec_cb_size, codebook_dim=cfg.dec_cb_dim, vector_quantization_hidden_dimension=cfg.dec_cb_dim,
                      hidden_size=cfg.dec_q_dim, num_quantizers=16, num_semantic_quantizers=1)
    def codebook(prefix):
        cb = MM.MimiEuclideanCodebook(mcfg)
        cb.embed_sum.copy_(W[prefix + "._codebook.embedding_sum"].reshape(cfg.dec_cb_size, cfg.dec_cb_dim))
        cb.cluster_usage.copy_(W[prefix + "._codebook.cluster_usage"])
        cb.initialized.fill_(1.0) if hasattr(cb, "initialized") else None
        if hasattr(cb, "_embed"):
            cb._embed = None
        return cb
    ct = torch.from_numpy(codes)                                   # [T][16]
    first = codebook("decoder.quantizer.rvq_first.vq.layers.0").decode(ct[:, 0][None])          # [1][T][CD]
    rest = torch.zeros_like(first)
    for i in range(15):
        rest = rest + codebook(f"decoder.quantizer.rvq_rest.vq.layers.{i}").decode(ct[:, i + 1][None])
    pf = W["decoder.quantizer.rvq_first.output_proj.weight"].reshape(cfg.dec_q_dim, cfg.dec_cb_dim, 1)
    pr = W["decoder.quantizer.rvq_rest.output_proj.weight"].reshape(cfg.dec_q_dim, cfg.dec_cb_dim, 1)
    quant = F.conv1d(first.transpose(1, 2), pf) + F.conv1d(rest.transpose(1, 2), pr)            # [1][Q][T]
    out["quant"] = quant[0].numpy()

    # ---- D2: pre_conv (CausalConvNet k = 3) ----
    pre = M.Qwen3OmniMoeCausalConvNet(cfg.dec_q_dim, cfg.dec_latent, 3)
    assert load(pre, W, "decoder.pre_conv.") == 2
    x = pre(quant)
    out["pre_conv"] = x[0].numpy()

    # ---- D3: pre-transformer = in-proj + Code2WavTransformerModel + out-proj ----
    tcfg = Qwen3OmniMoeCode2WavConfig(hidden_size=cfg.dec_hidden, num_hidden_layers=cfg.dec_layers, num_attention_heads=cfg.dec_heads,
                                      num_key_value_heads=cfg.dec_heads, head_dim=cfg.dec_head_dim, intermediate_size=cfg.dec_inter,
                                      rms_norm_eps=cfg.dec_eps, rope_theta=cfg.dec_theta, sliding_window=4096, attention_bias=False,
                                      max_position_embeddings=8000, hidden_act="silu", attention_dropout=0.0)
    tcfg._attn_implementation = "eager"
    tm = M.Qwen3OmniMoeCode2WavTransformerModel(tcfg).eval()
    n = load(tm, W, "decoder.pre_transformer.")
    assert n == cfg.dec_layers * 11 + 1, n
    h = F.linear(x.transpose(1, 2), W["decoder.pre_transformer.input_proj.weight"].reshape(cfg.dec_hidden, cfg.dec_latent),
                 W["decoder.pre_transformer.input_proj.bias"])
    h = tm(inputs_embeds=h).last_hidden_state
    h = F.linear(h, W["decoder.pre_transformer.output_proj.weight"].reshape(cfg.dec_latent, cfg.dec_hidden),
                 W["decoder.pre_transformer.output_proj.bias"])
    x = h.transpose(1, 2).contiguous()
    out["pre_transformer"] = x[0].numpy()

    # ---- D4-D9: upsample + decoder of a Code2Wav instance at the latent width ----
    ccfg = Qwen3OmniMoeCode2WavConfig(hidden_size=cfg.dec_latent, decoder_dim=cfg.dec_dim, upsample_rates=list(cfg.dec_up_rates),
                                      upsampling_ratios=list(cfg.dec_up_ratios), num_hidden_layers=1, num_attention_heads=2,
                                      num_key_value_heads=2, head_dim=cfg.dec_latent // 2, intermediate_size=16,
                                      codebook_size=cfg.dec_cb_size, num_quantizers=16)
    c2w = M.Qwen3OmniMoeCode2Wav(ccfg).eval()
    n_up = load(c2w.upsample, W, "decoder.upsample."); n_dec = load(c2w.decoder, W, "decoder.decoder.")
    assert n_up == 2 * 11 and n_dec == 2 + 4 * (2 + 2 + 3 * 8) + 2 + 2, (n_up, n_dec)
    for i, blocks in enumerate(c2w.upsample):
        for blk in blocks:
            x = blk(x)                                   # k = s: the two-sided trim is zero, identical to the reference
        out[f"up{i}"] = x[0].numpy()
    x = c2w.decoder[0](x)
    out["init"] = x[0].numpy()
    for b in range(4):
        blk = c2w.decoder[1 + b].block
        r = cfg.dec_up_rates[b]
        xin = blk[0](x)                                  # SnakeBeta
        # reference rule: full transposed conv, drop k - s samples on the right only (causal_trans_conv.rs:76-99)
        full = F.conv_transpose1d(xin, blk[1].conv.weight, blk[1].conv.bias, stride=r)
        y = full[..., : full.shape[-1] - r]
        if b == 0:
            out["hf_transconv_blk0"] = blk[1](xin)[0].numpy()      # the HF module's own (two-sided) trim, for the shift relation
            out["ref_rule_transconv_blk0"] = y[0].numpy()
            out["transconv_blk0_input"] = xin[0].numpy()
        x = y
        for u in range(3):
            x = blk[2 + u](x)                            # Code2WavDecoderResidualUnit (dilations 1, 3, 9)
        out[f"blk{b}"] = x[0].numpy()
    x = c2w.decoder[5](x)
    x = c2w.decoder[6](x)
    out["pcm"] = x.clamp(min=-1, max=1)[0, 0].numpy()
    out["pcm_preclamp"] = x[0, 0].numpy()

    # ---- A4 / A2: talker layers with HF's Qwen3 decoder layer (q/k-norm, GQA, RoPE, SwiGLU) on a prompt's prefill embeddings ----
    import oracle as O
    from common import oracle_model
    om = oracle_model(cfg, seed=SEED, which=1)
    utt = q.Utterance(synthetic_prompt(9, 3), q.Speaker.Ryan, q.Language.English, seed=1)
    osess = O.OracleSession(om, utt, q.SynthesisOptions(max_length=4, seed=1))
    emb = osess.prefill_embeds()                         # [S][H]: the oracle's prompt assembly is NOT under test here, only the layers
    osess.close(); om.close()
    lcfg = Qwen3OmniMoeTalkerCodePredictorConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                                                 num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim,
                                                 rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, attention_bias=False, hidden_act="silu",
                                                 attention_dropout=0.0, sliding_window=None, max_position_embeddings=4096)
    lcfg._attn_implementation = "eager"
    S = emb.shape[0]
    hcur = torch.from_numpy(emb)[None]
    pos = torch.arange(S)[None]
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
    fr = pos[..., None].float() * inv
    cos, sin = torch.cat([fr, fr], -1).cos(), torch.cat([fr, fr], -1).sin()
    mask = torch.full((1, 1, S, S), float("-inf")).triu(1)
    for i in range(cfg.n_layers):
        layer = M.Qwen3OmniMoeTalkerCodePredictorDecoderLayer(lcfg, i).eval()
        nl = load(layer, W, f"talker.model.layers.{i}.")
        assert nl == 11, nl
        hcur = layer(hcur, attention_mask=mask, position_ids=pos, position_embeddings=(cos, sin))
        hcur = hcur[0] if isinstance(hcur, tuple) else hcur
    norm = M.Qwen3OmniMoeRMSNorm(cfg.hidden, eps=cfg.rms_eps); norm.weight.copy_(W["talker.model.norm.weight"])
    hn = norm(hcur)[0, -1]
    out["talker_prefill_embeds"] = emb
    out["talker_last_hidden"] = hn.numpy()
    out["talker_logits"] = F.linear(hn, W["talker.codec_head.weight"].reshape(cfg.codec_vocab, cfg.hidden)).numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})
    mimi_fixture()


def cp_loop_fixture():
    """A3 (code_predictor.rs:320-416) against upstream's own code-predictor module. The Qwen3-Omni talker's code predictor IS
    this loop: forward(inputs_embeds = [last_hidden, layer-0 embed]) is the two-token first pass scored by lm_head[0]; every
    later pass embeds the previous code with codec_embedding[generation_steps - 1], runs the layers on the module's own KV
    cache at the next position, and scores with lm_head[generation_steps] (modeling_qwen3_omni_moe.py, forward()).
    Differences from /root/reference handled here, explicitly:
      * Qwen3-TTS 1.7B feeds the code predictor through `small_to_mtp_projection` (Linear 2048 -> 1024 + bias,
        code_predictor.rs:337-345, 386-396); Qwen3-Omni has no such layer. The fixture applies it OUTSIDE the module: the two
        prefill rows are projected with F.linear, and the 15 embedding tables handed to the module are the reference's tables
        with every row projected (a gather followed by a Linear == a gather from the projected table, bit for bit in f32).
      * upstream SAMPLES the sub-talker codes (docs/QWEN3_TTS_ARCHITECTURE.md:308-317); the reference takes the argmax
        (code_predictor.rs:374-375, 407-408). The loop below is greedy — only the choice rule differs, not the data flow.
    Written for two configurations: the tiny one and one with the production layer count (5) and a 2-way GQA."""
    import dataclasses
    import oracle as O
    from common import oracle_model
    res = {}
    cases = {"tiny": q.tiny(), "mid": dataclasses.replace(q.tiny(), hidden=128, inter=256, n_heads=2, n_kv_heads=1, cp_hidden=64, cp_inter=192, cp_layers=5, cp_heads=4, cp_kv_heads=2)}
    for tag, cfg in cases.items():
        W = checkpoint(cfg, SEED)
        ccfg = Qwen3OmniMoeTalkerCodePredictorConfig(vocab_size=cfg.cp_vocab, hidden_size=cfg.cp_hidden, intermediate_size=cfg.cp_inter, num_hidden_layers=cfg.cp_layers,
                                                     num_attention_heads=cfg.cp_heads, num_key_value_heads=cfg.cp_kv_heads, head_dim=cfg.head_dim, hidden_act="silu",
                                                     rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, attention_bias=False, attention_dropout=0.0,
                                                     sliding_window=None, max_position_embeddings=64, num_code_groups=cfg.n_groups, use_cache=True)
        ccfg._attn_implementation = "eager"
        cp = M.Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration(ccfg).eval()
        sd = cp.state_dict(); got = 0
        pw = W["talker.code_predictor.small_to_mtp_projection.weight"].reshape(cfg.cp_hidden, cfg.hidden)
        pb = W["talker.code_predictor.small_to_mtp_projection.bias"]
        for k, v in sd.items():
            if k.startswith("model.layers.") or k == "model.norm.weight":
                name = "talker.code_predictor." + k
                sd[k] = W[name].reshape(v.shape).clone(); got += 1
            elif k.startswith("lm_head."):
                sd[k] = W["talker.code_predictor." + k].reshape(v.shape).clone(); got += 1
            elif k.startswith("model.codec_embedding."):
                g = int(k.split(".")[2])
                tab = W[f"talker.code_predictor.model.codec_embedding.{g}.weight"].reshape(cfg.cp_vocab, cfg.hidden)
                sd[k] = F.linear(tab, pw, pb).clone(); got += 1            # every row through small_to_mtp_projection
            else:
                raise SystemExit(f"unexpected parameter {k}")
        assert got == len(sd), (got, len(sd))
        cp.load_state_dict(sd)
        rng = np.random.default_rng(31 + len(tag))
        B = 3
        last_hidden = rng.standard_normal((B, cfg.hidden)).astype(np.float32)
        sem_tok = rng.integers(0, 2048, size=B)
        sem_embed = W["talker.model.codec_embedding.weight"].reshape(cfg.codec_vocab, cfg.hidden)[sem_tok].numpy()
        ids = np.zeros((B, 15), np.int64); logits = np.zeros((B, 15, cfg.cp_vocab), np.float32)
        for b in range(B):
            x = F.linear(torch.from_numpy(np.stack([last_hidden[b], sem_embed[b]]))[None], pw, pb)       # [1, 2, cp_hidden]
            o = cp(inputs_embeds=x, use_cache=True)
            assert o.generation_steps == 1
            lg = o.logits[0, -1]; tok = int(lg.argmax()); ids[b, 0] = tok; logits[b, 0] = lg.numpy()
            past = o.past_key_values; step = o.generation_steps
            for g in range(1, 15):
                o = cp(input_ids=torch.tensor([[tok]]), past_key_values=past, use_cache=True, generation_steps=step)
                lg = o.logits[0, -1]; tok = int(lg.argmax()); ids[b, g] = tok; logits[b, g] = lg.numpy()
                past = o.past_key_values; step = o.generation_steps
            assert step == 15
        # the same inputs through the C oracle's q3o_session_cp_generate are compared in tests/test_oracle_vs_hf.py; here only a
        # sanity print so that a broken fixture is noticed when it is made
        om = oracle_model(cfg, seed=SEED, which=1)
        osess = O.OracleSession(om, q.Utterance(synthetic_prompt(5, 1), q.Speaker.Ryan, q.Language.English, seed=1), q.SynthesisOptions(max_length=2, seed=1))
        agree = 0
        for b in range(B):
            oc, ol = osess.cp_generate(last_hidden[b], sem_embed[b])
            agree += int((np.asarray(oc) == ids[b]).all())
            print(f"[cp_loop {tag}] row {b}: ids equal {bool((np.asarray(oc) == ids[b]).all())}, max |logit diff| {np.abs(ol - logits[b]).max():.2e}")
        osess.close(); om.close()
        st = np.sort(logits.astype(np.float64), axis=2)
        res.update({f"{tag}_last_hidden": last_hidden, f"{tag}_sem_embed": sem_embed, f"{tag}_ids": ids.astype(np.uint32), f"{tag}_logits": logits,
                    f"{tag}_top2_margin": (st[..., -1] - st[..., -2]).astype(np.float32),
                    f"{tag}_cfg": np.array([cfg.hidden, cfg.inter, cfg.n_heads, cfg.n_kv_heads, cfg.cp_hidden, cfg.cp_inter, cfg.cp_layers, cfg.cp_heads, cfg.cp_kv_heads], np.int32)})
    path = os.path.join(os.path.dirname(OUT), "hf_cp_loop.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, {k: v.shape for k, v in res.items()})


def frame_loop_fixture():
    """A1 + A10 (lib.rs:530-656: the frame loop — sample code 0 from the talker's logits, run the code predictor on
    [last hidden, embed(code 0)], feed the talker the SUM of the sixteen code embeddings plus the next trailing-text row, or the
    tts_pad embedding once the text has run out) against upstream's own loop: Qwen3OmniMoeTalkerForConditionalGeneration.generate()
    -> prepare_inputs_for_generation() (modeling_qwen3_omni_moe.py) on the module's own KV caches, its own
    code_predictor.generate() inside, greedy. Until round 6 this loop was only ever compared with tests/np_reference.py (same author).
    What is NOT under test: the prompt assembly (the oracle's prefill embeddings, trailing rows and pad row are the loop's inputs).
    Differences from /root/reference handled here, explicitly:
      * the Qwen3-Omni talker is a sparse MoE; Qwen3-TTS's is dense. Every layer's `mlp` is replaced by upstream's own dense SwiGLU
        module (Qwen3OmniMoeTalkerTextMLP) before the weights are loaded — the loop, the attention, the norms, the caches, the rotary
        embedding (3-axis interleaved M-RoPE, which with the text-only positions used here IS plain RoPE) stay upstream's;
      * upstream's code predictor has the talker's width: the fixture uses the 0.6B topology (no small_to_mtp_projection;
        `tiny_same_width`), where the reference's loop is exactly upstream's (lib.rs sums the UNprojected embeddings in either case);
      * upstream samples; the loop is run greedy on both sides, once with the plain logits and once with the reference's default
        penalties carried through the loop (repetition penalty 1.05, min_new_tokens 2 with the EOS id live; with these weights the
        greedy path never revisits a token, so they ride along without changing a choice — the rules themselves are held to
        upstream's processors by hf_sampling.npz) — upstream's RepetitionPenalty /
        SuppressTokens / MinNewTokens logits processors against q3o_apply_penalties (lib.rs:1271-1322);
      * upstream's standalone talker config lacks `spatial_merge_size` (read by its __init__, only used for vision inputs): set by hand.
    Two configurations: the tiny one and one with three talker layers, 2-way GQA and the production code-predictor depth."""
    import dataclasses
    import oracle as O
    from common import oracle_model
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeTalkerConfig
    res = {}
    base = q.tiny_same_width()
    cases = {"tiny": base,
             "mid": dataclasses.replace(base, hidden=128, inter=256, n_layers=3, n_heads=4, n_kv_heads=2, cp_hidden=128, cp_inter=192, cp_layers=5, cp_heads=4, cp_kv_heads=2)}
    runs = {"plain": dict(repetition_penalty=1.0, min_new_tokens=0, eos_token_id=None),
            "penalties": dict(repetition_penalty=1.05, min_new_tokens=2, eos_token_id=q.api.CODEC_EOS_TOKEN_ID)}
    N = 24
    for tag, cfg in cases.items():
        W = checkpoint(cfg, SEED)
        hd6 = cfg.head_dim // 6
        text = dict(vocab_size=cfg.codec_vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                    num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_eps,
                    rope_parameters=dict(rope_type="default", rope_theta=cfg.rope_theta, mrope_section=[cfg.head_dim // 2 - 2 * hd6, hd6, hd6], interleaved=True),
                    moe_intermediate_size=8, shared_expert_intermediate_size=cfg.inter, num_experts=2, num_experts_per_tok=1, max_position_embeddings=4096)
        cp = dict(vocab_size=cfg.cp_vocab, hidden_size=cfg.cp_hidden, intermediate_size=cfg.cp_inter, num_hidden_layers=cfg.cp_layers,
                  num_attention_heads=cfg.cp_heads, num_key_value_heads=cfg.cp_kv_heads, head_dim=cfg.head_dim, hidden_act="silu",
                  rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, attention_bias=False, attention_dropout=0.0, sliding_window=None,
                  max_position_embeddings=64, num_code_groups=cfg.n_groups, use_cache=True)
        tc = Qwen3OmniMoeTalkerConfig(text_config=text, code_predictor_config=cp, num_code_groups=cfg.n_groups, thinker_hidden_size=cfg.hidden)
        tc.spatial_merge_size = 2
        tc._attn_implementation = "eager"; tc.text_config._attn_implementation = "eager"; tc.code_predictor_config._attn_implementation = "eager"
        talker = M.Qwen3OmniMoeTalkerForConditionalGeneration(tc)
        for layer in talker.model.layers:
            layer.mlp = M.Qwen3OmniMoeTalkerTextMLP(tc.text_config, intermediate_size=cfg.inter)
        talker.eval()
        sd = talker.state_dict(); got = 0
        for k, v in sd.items():
            if k.startswith(("text_projection.", "hidden_projection.")):
                continue                                  # thinker -> talker adapters of Qwen3-Omni: outside the talker's forward, unused here
            name = "talker." + k
            assert name in W and W[name].numel() == v.numel(), (name, tuple(v.shape))
            sd[k] = W[name].reshape(v.shape).clone(); got += 1
        assert got == len(sd) - 8, (got, len(sd))
        talker.load_state_dict(sd)
        om = oracle_model(cfg, seed=SEED, which=1)
        utt = q.Utterance(synthetic_prompt(9, 3), q.Speaker.Ryan, q.Language.English, seed=1)
        for rn, ro in runs.items():
            osess = O.OracleSession(om, utt, q.SynthesisOptions(max_length=N, seed=1, temperature=0.0, **ro))
            emb = osess.prefill_embeds(); tr, pad = osess.trailing()
            assert 0 < tr.shape[0] < N - 2                 # both branches of the text / pad choice are walked
            S, V = emb.shape[0], cfg.codec_vocab
            sup = [t for t in range(V - 1024, V) if t != q.api.CODEC_EOS_TOKEN_ID]        # generation/tts.rs:21-43
            kw = dict(repetition_penalty=ro["repetition_penalty"]) if ro["repetition_penalty"] != 1.0 else {}
            if ro["eos_token_id"] is not None:
                kw.update(eos_token_id=ro["eos_token_id"], min_new_tokens=ro["min_new_tokens"])
            talker.rope_deltas = None
            out = talker.generate(inputs_embeds=torch.from_numpy(emb)[None], attention_mask=torch.ones(1, S, dtype=torch.long),
                                  talker_input_ids=torch.zeros(1, S, dtype=torch.long), trailing_text_hidden=torch.from_numpy(tr)[None],
                                  tts_pad_embed=torch.from_numpy(pad)[None, None], max_new_tokens=N, do_sample=False, suppress_tokens=sup,
                                  output_hidden_states=True, output_logits=True, return_dict_in_generate=True, use_cache=True, pad_token_id=0, **kw)
            code0 = out.sequences[0].numpy().astype(np.uint32)
            resid = np.stack([h[1][0].numpy() for h in out.hidden_states[1:]]).astype(np.uint32)     # step i carries frame i - 1's sixteen codes
            logits = np.stack([l[0].numpy() for l in out.logits]).astype(np.float32)                 # raw talker logits of every step
            assert (resid[:, 0] == code0[:resid.shape[0]]).all()
            oc, otl, ocl = osess.generate(capture=True)
            n = min(len(oc), len(code0))
            print(f"[frame_loop {tag}/{rn}] upstream {len(code0)} frames, oracle {len(oc)}; code 0 equal {bool((oc[:n, 0] == code0[:n]).all())}, "
                  f"all 16 codes of the first {resid.shape[0]} frames equal {bool((oc[:resid.shape[0]] == resid[:len(oc)]).all())}, "
                  f"max |talker logit diff| {np.abs(otl[:n] - logits[:n]).max():.2e}")
            osess.close()
            st = np.sort(logits.astype(np.float64), axis=1)
            res.update({f"{tag}_{rn}_prefill_embeds": emb, f"{tag}_{rn}_trailing": tr, f"{tag}_{rn}_pad": pad, f"{tag}_{rn}_code0": code0,
                        f"{tag}_{rn}_codes": resid, f"{tag}_{rn}_top2_margin": (st[:, -1] - st[:, -2]).astype(np.float32)})
            if rn == "plain":
                res[f"{tag}_{rn}_logits"] = logits
        om.close()
        res[f"{tag}_cfg"] = np.array([cfg.hidden, cfg.inter, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.cp_hidden, cfg.cp_inter, cfg.cp_layers, cfg.cp_heads, cfg.cp_kv_heads], np.int32)
    path = os.path.join(os.path.dirname(OUT), "hf_frame_loop.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, {k: v.shape for k, v in res.items()})


def test_clip(n, seed):
    """deterministic speech-like test signal in [-0.6, 0.6]"""
    t = np.arange(n) / 24000.0
    rng = np.random.default_rng(seed)
    x = 0.3 * np.sin(2 * np.pi * 180.0 * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 2.5 * t)) + 0.15 * np.sin(2 * np.pi * 1250.0 * t + 0.7) + \
        0.05 * rng.standard_normal(n)
    return x.astype(np.float32)


def mimi_fixture():
    """ICL reference-audio encoder: the repo's seeded synthetic speech-encoder checkpoint loaded into HF's MimiModel
    (transformers.models.mimi), stage taps + codes for a tiny and the full-size configuration."""
    from qwen3_tts_rs_amd.speech_encoder import SpeechEncoder, SpeechEncoderConfig, tiny_speech_config, synthetic_speech_checkpoint
    res = {}
    for tag, scfg, n in (("tiny", tiny_speech_config(), 2000), ("full", SpeechEncoderConfig(), 30000)):
        enc = SpeechEncoder(scfg, device=-1)
        W = {name: torch.from_numpy(arr.copy()) for name, arr in synthetic_speech_checkpoint(enc, SEED)}
        enc.close()
        prod = int(np.prod(scfg.ratios))
        mcfg = MimiConfig(sampling_rate=24000, frame_rate=24000 / prod / 2, audio_channels=1, hidden_size=scfg.hidden, num_filters=scfg.n_filters,
                          num_residual_layers=1, upsampling_ratios=list(reversed(scfg.ratios)), kernel_size=scfg.kernel, last_kernel_size=scfg.last_kernel,
                          residual_kernel_size=scfg.res_kernel, dilation_growth_rate=2, use_causal_conv=True, pad_mode="constant", compress=scfg.compress,
                          codebook_size=scfg.cb_size, codebook_dim=scfg.cb_dim, vector_quantization_hidden_dimension=scfg.cb_dim,
                          num_quantizers=scfg.n_q, num_semantic_quantizers=scfg.n_sem, num_hidden_layers=scfg.n_layers, intermediate_size=scfg.inter,
                          num_attention_heads=scfg.n_heads, num_key_value_heads=scfg.n_heads, head_dim=scfg.head_dim, hidden_act="gelu",
                          norm_eps=scfg.norm_eps, rope_theta=scfg.rope_theta, sliding_window=scfg.window, attention_bias=False, use_conv_shortcut=False,
                          upsample_groups=scfg.hidden)
        mcfg._attn_implementation = "eager"
        mm = MM.MimiModel(mcfg).eval()
        sd = mm.state_dict(); got = 0
        for k in sd:
            name = "encoder." + k
            if name in W:
                assert W[name].numel() == sd[k].numel(), (name, W[name].numel(), tuple(sd[k].shape))
                sd[k] = W[name].reshape(sd[k].shape).clone(); got += 1
        assert got == len(W), (got, len(W))
        mm.load_state_dict(sd)
        for m_ in mm.modules():
            if isinstance(m_, MM.MimiEuclideanCodebook):
                m_._embed = None
        x = torch.from_numpy(test_clip(n, 7))[None, None]
        emb = mm.encoder(x)
        tr = mm.encoder_transformer(emb.transpose(1, 2))[0].transpose(1, 2)
        ds = mm.downsample(tr)
        codes = mm.quantizer.encode(ds, scfg.n_q)                                # [K][1][T]
        codes2 = mm.encode(x, num_quantizers=scfg.n_q).audio_codes                # the public entry point agrees
        assert torch.equal(codes[:, 0], codes2[0])
        res[f"mimi_{tag}_n"] = np.array([n])
        res[f"mimi_{tag}_seanet"] = emb[0].numpy(); res[f"mimi_{tag}_transformer"] = tr[0].numpy(); res[f"mimi_{tag}_downsample"] = ds[0].numpy()
        res[f"mimi_{tag}_codes"] = codes[:, 0].transpose(0, 1).numpy().astype(np.uint32)        # [T][n_q]
        print(tag, {k: v.shape for k, v in res.items() if tag in k})
    path = os.path.join(os.path.dirname(OUT), "hf_mimi.npz")
    np.savez_compressed(path, **res)
    print("wrote", path)


def sampling_fixture():
    """Sampler filters (A9: lib.rs:1271-1322 penalties, sampling.rs:189-262 top-k / top-p) against Hugging Face's own
    logits processors (transformers.generation.logits_process): RepetitionPenaltyLogitsProcessor, TopKLogitsWarper,
    TopPLogitsWarper, TemperatureLogitsWarper. Same published rules; the one structural difference: HF's top-p walks the
    ASCENDING sort and removes the tail whose cumulative probability is <= 1 - p, the reference walks the DESCENDING sort
    and keeps tokens until the cumulative probability exceeds p — the same kept set in exact arithmetic, decided by
    different f32 sums, so a boundary token whose inclusion hangs on the last ulps may differ (the test reports how many
    of the cases have such a token; none of the seeded cases does)."""
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TopKLogitsWarper, TopPLogitsWarper,
                                                        TemperatureLogitsWarper)
    rng = np.random.default_rng(7)
    V = 3072
    cases = []
    for i in range(16):
        scale = [1.0, 3.0, 8.0][i % 3]
        logits = (rng.standard_normal(V) * scale).astype(np.float32)
        n_seen = int(rng.integers(0, 60))
        seen_ids = rng.choice(V - 1024, size=n_seen, replace=False).astype(np.int64)
        pen = [1.0, 1.05, 1.3][i % 3]; k = [50, 30, 5, 0][i % 4]; p = [0.9, 0.8, 0.95, 1.0][(i // 2) % 4]; temp = [0.9, 1.0, 0.7][(i // 3) % 3]
        x = torch.from_numpy(logits.copy())[None]
        ids = torch.from_numpy(seen_ids)[None]
        after_pen = RepetitionPenaltyLogitsProcessor(pen)(ids, x.clone()) if (pen != 1.0 and n_seen) else x.clone()
        after_t = TemperatureLogitsWarper(temp)(None, after_pen.clone()) if temp != 1.0 else after_pen.clone()
        after_k = TopKLogitsWarper(k)(None, after_t.clone()) if k > 0 else after_t.clone()
        after_p = TopPLogitsWarper(p)(None, after_k.clone()) if p < 1.0 else after_k.clone()
        cases.append(dict(logits=logits, seen=seen_ids, pen=pen, k=k, p=p, temp=temp, after_pen=after_pen[0].numpy(),
                          after_t=after_t[0].numpy(), keep_k=np.isfinite(after_k[0].numpy()), keep_p=np.isfinite(after_p[0].numpy())))
    res = {"n": np.int64(len(cases))}
    for i, c in enumerate(cases):
        for kk, v in c.items():
            res[f"c{i}_{kk}"] = np.asarray(v)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_sampling.npz")
    np.savez_compressed(path, **res)
    print("wrote", path)


if __name__ == "__main__":
    if "mimi" in sys.argv[1:]:
        mimi_fixture()
    elif "cp_loop" in sys.argv[1:]:
        cp_loop_fixture()
    elif "frame_loop" in sys.argv[1:]:
        frame_loop_fixture()
    elif "sampling" in sys.argv[1:]:
        sampling_fixture()
    else:
        main()
