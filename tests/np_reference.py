"""Independent numpy (float64) restatement of the reference hot path, written from the reference
sources (not from oracle/q3_oracle.c) to cross-check the C oracle. TEST INFRASTRUCTURE.

Reference files: src/models/transformer.rs:21-69,154-181,247-467; src/models/talker.rs:316-320,451-491,716-736,
823-841; src/models/code_predictor.rs:320-416; src/lib.rs:508-519,612-622; src/models/codec/*.rs.
Everything is float64, so agreement with the f32 oracle is expected to ~1e-5 relative.
"""
import numpy as np

IM_START, ASSISTANT, NEWLINE = 151644, 77091, 198
TTS_PAD, TTS_BOS, TTS_EOS = 151671, 151672, 151673
CODEC_PAD, CODEC_BOS, CODEC_EOS, CODEC_THINK, CODEC_THINK_BOS, CODEC_THINK_EOS = 2148, 2149, 2150, 2154, 2156, 2157


class W:
    """name → float64 array with shape lookup."""

    def __init__(self):
        self.t = {}

    def add(self, name, arr, dtype):
        if dtype == 1:
            a = (np.asarray(arr, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
        else:
            a = np.asarray(arr, dtype=np.float32)
        self.t[name] = a.astype(np.float64)

    def g(self, name, *shape):
        return self.t[name].reshape(shape)


def rms_norm(x, w, eps):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * w


def silu(x):
    return x / (1.0 + np.exp(-x))


def rope_cos_sin(theta, hd, positions):
    inv = 1.0 / (np.float32(theta) ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd)))   # f32 powf like the reference
    f = (np.asarray(positions, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float32)
    return np.cos(f.astype(np.float64)), np.sin(f.astype(np.float64))


def rotate_half(x, cos, sin):
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h], x[..., h:]
    return np.concatenate([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1)


def decoder_layer(w, p, x, cache, offset, H, I, nh, nkv, hd, eps, theta):
    """x [S,H]; cache dict with 'k','v' lists of [nkv,hd] rows."""
    S = x.shape[0]
    h1 = rms_norm(x, w.g(p + ".input_layernorm.weight", H), eps)
    q = (h1 @ w.g(p + ".self_attn.q_proj.weight", nh * hd, H).T).reshape(S, nh, hd)
    k = (h1 @ w.g(p + ".self_attn.k_proj.weight", nkv * hd, H).T).reshape(S, nkv, hd)
    v = (h1 @ w.g(p + ".self_attn.v_proj.weight", nkv * hd, H).T).reshape(S, nkv, hd)
    q = rms_norm(q, w.g(p + ".self_attn.q_norm.weight", hd), eps)
    k = rms_norm(k, w.g(p + ".self_attn.k_norm.weight", hd), eps)
    cos, sin = rope_cos_sin(theta, hd, np.arange(offset, offset + S))
    q = rotate_half(q, cos[:, None, :], sin[:, None, :])
    k = rotate_half(k, cos[:, None, :], sin[:, None, :])
    for s in range(S):
        cache["k"].append(k[s]); cache["v"].append(v[s])
    K = np.stack(cache["k"]); V = np.stack(cache["v"])       # [L, nkv, hd]
    n_rep = nh // nkv
    out = np.zeros((S, nh, hd))
    for s in range(S):
        L = offset + s + 1
        for h in range(nh):
            sc = (K[:L, h // n_rep] @ q[s, h]) * (1.0 / np.sqrt(hd))
            pr = np.exp(sc - sc.max()); pr /= pr.sum()
            out[s, h] = pr @ V[:L, h // n_rep]
    ao = out.reshape(S, nh * hd) @ w.g(p + ".self_attn.o_proj.weight", H, nh * hd).T
    summ = ao + x
    nrm = rms_norm(summ, w.g(p + ".post_attention_layernorm.weight", H), eps)
    g = nrm @ w.g(p + ".mlp.gate_proj.weight", I, H).T
    u = nrm @ w.g(p + ".mlp.up_proj.weight", I, H).T
    return summ + (silu(g) * u) @ w.g(p + ".mlp.down_proj.weight", H, I).T


class NpModel:
    def __init__(self, cfg, w: W):
        self.c = cfg; self.w = w

    def text_proj(self, ids):
        c, w = self.c, self.w
        e = w.g("talker.model.text_embedding.weight", c.text_vocab, c.text_dim)[np.asarray(ids, dtype=np.int64)]
        h = silu(e @ w.g("talker.text_projection.linear_fc1.weight", c.text_dim, c.text_dim).T + w.g("talker.text_projection.linear_fc1.bias", c.text_dim))
        return h @ w.g("talker.text_projection.linear_fc2.weight", c.hidden, c.text_dim).T + w.g("talker.text_projection.linear_fc2.bias", c.hidden)

    def codec_emb(self, ids):
        return self.w.g("talker.model.codec_embedding.weight", self.c.codec_vocab, self.c.hidden)[np.asarray(ids, dtype=np.int64)]

    def prefill_custom_voice(self, text_ids, speaker_id, language_id):
        role = self.text_proj([IM_START, ASSISTANT, NEWLINE])
        cod = self.codec_emb([CODEC_THINK, CODEC_THINK_BOS, language_id, CODEC_THINK_EOS, speaker_id, CODEC_PAD, CODEC_BOS])
        pad = self.text_proj([TTS_PAD]); bos = self.text_proj([TTS_BOS])
        overlay = np.concatenate([np.repeat(pad, 5, 0), bos], 0) + cod[:6]
        hid = np.concatenate([role, overlay], 0)
        if len(text_ids) > 0:
            hid = np.concatenate([hid, self.text_proj([text_ids[0]]) + cod[6:7]], 0)
        return hid

    def talker_layers(self, x, caches, offset):
        c = self.c
        for i in range(c.n_layers):
            x = decoder_layer(self.w, f"talker.model.layers.{i}", x, caches[i], offset, c.hidden, c.inter, c.n_heads,
                              c.n_kv_heads, c.head_dim, c.rms_eps, c.rope_theta)
        return x

    def talker_head(self, x):
        c = self.c
        n = rms_norm(x, self.w.g("talker.model.norm.weight", c.hidden), c.rms_eps)
        return n, n[-1:] @ self.w.g("talker.codec_head.weight", c.codec_vocab, c.hidden).T

    def cp_generate(self, last_hidden, sem_embed):
        c, w = self.c, self.w
        H, CH = c.hidden, c.cp_hidden
        caches = [{"k": [], "v": []} for _ in range(c.cp_layers)]

        def proj(x):
            if H != CH:
                return x @ w.g("talker.code_predictor.small_to_mtp_projection.weight", CH, H).T + w.g("talker.code_predictor.small_to_mtp_projection.bias", CH)
            return x

        def layers(x, off):
            for i in range(c.cp_layers):
                x = decoder_layer(w, f"talker.code_predictor.model.layers.{i}", x, caches[i], off, CH, c.cp_inter, c.cp_heads,
                                  c.cp_kv_heads, c.head_dim, c.rms_eps, c.rope_theta)
            return rms_norm(x, w.g("talker.code_predictor.model.norm.weight", CH), c.rms_eps)

        x = layers(proj(np.stack([last_hidden, sem_embed])), 0)
        logits = [x[-1] @ w.g("talker.code_predictor.lm_head.0.weight", c.cp_vocab, CH).T]
        codes = [int(np.argmax(logits[0]))]
        off = 2
        for g in range(1, 15):
            e = w.g(f"talker.code_predictor.model.codec_embedding.{g - 1}.weight", c.cp_vocab, H)[codes[-1]]
            x = layers(proj(e[None]), off)
            logits.append(x[-1] @ w.g(f"talker.code_predictor.lm_head.{g}.weight", c.cp_vocab, CH).T)
            codes.append(int(np.argmax(logits[-1])))
            off += 1
        return codes, np.stack(logits)

    # ---------------- codec decoder ----------------
    def decode(self, codes):
        """codes [T,16] → pcm [1920 T] (decoder_12hz.rs:411-505)."""
        c, w = self.c, self.w
        T = codes.shape[0]
        CB, CD, Q, LAT, DH = c.dec_cb_size, c.dec_cb_dim, c.dec_q_dim, c.dec_latent, c.dec_hidden

        def cb(p):
            es = w.g(p + "._codebook.embedding_sum", CB, CD); us = np.maximum(w.g(p + "._codebook.cluster_usage", CB), np.float64(np.float32(1e-7)))
            return es / us[:, None]
        first = cb("decoder.quantizer.rvq_first.vq.layers.0")[codes[:, 0].astype(np.int64) % CB]
        rest = np.zeros((T, CD))
        for i in range(15):
            rest = rest + cb(f"decoder.quantizer.rvq_rest.vq.layers.{i}")[codes[:, i + 1].astype(np.int64)]
        qz = first @ w.g("decoder.quantizer.rvq_first.output_proj.weight", Q, CD).T + rest @ w.g("decoder.quantizer.rvq_rest.output_proj.weight", Q, CD).T
        x = qz.T                                              # [Q,T]
        x = causal_conv(x, w.g("decoder.pre_conv.conv.weight", LAT, Q, 3), w.g("decoder.pre_conv.conv.bias", LAT), 1)
        h = x.T @ w.g("decoder.pre_transformer.input_proj.weight", DH, LAT).T + w.g("decoder.pre_transformer.input_proj.bias", DH)
        nh, hd, DI = c.dec_heads, c.dec_head_dim, c.dec_inter
        cos, sin = rope_cos_sin(c.dec_theta, hd, np.arange(T))
        mask = np.triu(np.full((T, T), -np.inf), 1)
        for l in range(c.dec_layers):
            p = f"decoder.pre_transformer.layers.{l}"
            n = rms_norm(h, w.g(p + ".input_layernorm.weight", DH), c.dec_eps)
            qq = (n @ w.g(p + ".self_attn.q_proj.weight", nh * hd, DH).T).reshape(T, nh, hd)
            kk = (n @ w.g(p + ".self_attn.k_proj.weight", nh * hd, DH).T).reshape(T, nh, hd)
            vv = (n @ w.g(p + ".self_attn.v_proj.weight", nh * hd, DH).T).reshape(T, nh, hd)
            qq = rotate_half(qq, cos[:, None, :], sin[:, None, :]); kk = rotate_half(kk, cos[:, None, :], sin[:, None, :])
            att = np.einsum("ihd,jhd->hij", qq, kk) * hd ** -0.5 + mask[None]
            att = np.exp(att - att.max(-1, keepdims=True)); att /= att.sum(-1, keepdims=True)
            ao = np.einsum("hij,jhd->ihd", att, vv).reshape(T, nh * hd) @ w.g(p + ".self_attn.o_proj.weight", DH, nh * hd).T
            h = h + ao * w.g(p + ".self_attn_layer_scale.scale", DH)
            n = rms_norm(h, w.g(p + ".post_attention_layernorm.weight", DH), c.dec_eps)
            m = (silu(n @ w.g(p + ".mlp.gate_proj.weight", DI, DH).T) * (n @ w.g(p + ".mlp.up_proj.weight", DI, DH).T)) @ w.g(p + ".mlp.down_proj.weight", DH, DI).T
            h = h + m * w.g(p + ".mlp_layer_scale.scale", DH)
        h = rms_norm(h, w.g("decoder.pre_transformer.norm.weight", DH), c.dec_eps)
        x = (h @ w.g("decoder.pre_transformer.output_proj.weight", LAT, DH).T + w.g("decoder.pre_transformer.output_proj.bias", LAT)).T
        for i, r in enumerate(c.dec_up_ratios):
            p = f"decoder.upsample.{i}"
            x = causal_trans_conv(x, w.g(p + ".0.conv.weight", LAT, LAT, r), w.g(p + ".0.conv.bias", LAT), r)
            d = causal_conv(x, w.g(p + ".1.dwconv.conv.weight", LAT, 1, 7), w.g(p + ".1.dwconv.conv.bias", LAT), 1, groups=LAT).T
            mu = d.mean(-1, keepdims=True); var = ((d - mu) ** 2).mean(-1, keepdims=True)
            d = (d - mu) / np.sqrt(var + 1e-6) * w.g(p + ".1.norm.weight", LAT) + w.g(p + ".1.norm.bias", LAT)
            d = d @ w.g(p + ".1.pwconv1.weight", 4 * LAT, LAT).T + w.g(p + ".1.pwconv1.bias", 4 * LAT)
            d = 0.5 * d * (1.0 + erf_vec(d / np.sqrt(2.0)))
            d = (d @ w.g(p + ".1.pwconv2.weight", LAT, 4 * LAT).T + w.g(p + ".1.pwconv2.bias", LAT)) * w.g(p + ".1.gamma", LAT)
            x = x + d.T
        C = c.dec_dim
        x = causal_conv(x, w.g("decoder.decoder.0.conv.weight", C, LAT, 7), w.g("decoder.decoder.0.conv.bias", C), 1)
        for b, r in enumerate(c.dec_up_rates):
            p = f"decoder.decoder.{b + 1}.block"
            Co = C // 2
            x = snake(x, w.g(p + ".0.alpha", C), w.g(p + ".0.beta", C))
            x = causal_trans_conv(x, w.g(p + ".1.conv.weight", C, Co, 2 * r), w.g(p + ".1.conv.bias", Co), r)
            for u, dil in enumerate((1, 3, 9)):
                pu = f"{p}.{u + 2}"
                y = snake(x, w.g(pu + ".act1.alpha", Co), w.g(pu + ".act1.beta", Co))
                y = causal_conv(y, w.g(pu + ".conv1.conv.weight", Co, Co, 7), w.g(pu + ".conv1.conv.bias", Co), dil)
                y = snake(y, w.g(pu + ".act2.alpha", Co), w.g(pu + ".act2.beta", Co))
                y = causal_conv(y, w.g(pu + ".conv2.conv.weight", Co, Co, 1), w.g(pu + ".conv2.conv.bias", Co), 1)
                x = x + y
            C = Co
        x = snake(x, w.g("decoder.decoder.5.alpha", C), w.g("decoder.decoder.5.beta", C))
        x = causal_conv(x, w.g("decoder.decoder.6.conv.weight", 1, C, 7), w.g("decoder.decoder.6.conv.bias", 1), 1)
        return np.clip(x[0], -1.0, 1.0)


def erf_vec(x):
    from math import erf
    return np.vectorize(erf)(x)


def snake(x, alpha, beta):
    a = np.exp(alpha)[:, None]; b = np.exp(beta)[:, None]
    return x + np.sin(x * a) ** 2 / (b + 1e-9)


def causal_conv(x, w, b, dil, groups=1):
    """x [Cin,L], w [Cout,Cin/groups,K] — left pad dil*(K-1) (causal_conv.rs:94-103)."""
    cout, cing, K = w.shape
    cin, L = x.shape
    pad = dil * (K - 1)
    xp = np.concatenate([np.zeros((cin, pad)), x], 1)
    y = np.zeros((cout, L))
    cog = cout // groups
    for g in range(groups):
        xs = xp[g * cing:(g + 1) * cing]
        for k in range(K):
            y[g * cog:(g + 1) * cog] += w[g * cog:(g + 1) * cog, :, k] @ xs[:, k * dil:k * dil + L]
    return y + b[:, None]


def causal_trans_conv(x, w, b, stride):
    """x [Cin,L], w [Cin,Cout,K] (causal_trans_conv.rs:88-100): full transposed conv then right-trim K-stride."""
    cin, cout, K = w.shape
    L = x.shape[1]
    full = np.zeros((cout, (L - 1) * stride + K))
    for k in range(K):
        full[:, k:k + (L - 1) * stride + 1:stride] += w[:, :, k].T @ x
    trim = max(K - stride, 0)
    out = full[:, :full.shape[1] - trim] if trim > 0 else full
    return out + b[:, None]
