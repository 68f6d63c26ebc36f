/* q3_oracle_mimi.c — CPU restatement of the speech-tokenizer ENCODER (reference: src/models/codec/encoder_12hz.rs:34-144,
 * called by create_voice_clone_prompt, src/lib.rs:1172-1178, to turn the ICL reference audio into 16-codebook frames).
 * TEST INFRASTRUCTURE ONLY (see q3_oracle.h): loaded by tests/, never by the product.
 *
 * The reference does not implement this model itself: encoder_12hz.rs:23 instantiates candle-transformers' `mimi` modules
 * (crate candle-transformers 0.9, Cargo.toml:35, un-vendored and absent from /root/reference) with `mimi::Config::v0_1(Some(16))`
 * over the HF-format keys `encoder.*` of speech_tokenizer/model.safetensors. What is restated here is therefore the PUBLISHED
 * Mimi encoder algorithm for that checkpoint format, as written in Hugging Face transformers' models/mimi/modeling_mimi.py
 * (the implementation the checkpoint was exported for; encoder_12hz.rs:6 "a standard HuggingFace Mimi model"):
 *   MimiConv1d (causal left padding k_eff - stride, right padding up to a whole frame, "constant" / "replicate" modes),
 *   MimiEncoder (SEANet: conv7, 4 x {ResnetBlock(ELU, conv3, ELU, conv1) + ELU + strided conv 2r/r}, ELU, conv3),
 *   MimiTransformerLayer x 8 (LayerNorm, MHA with half-split RoPE and a 250-frame causal window, LayerScale, GELU MLP),
 *   downsample conv k=4 s=2 (replicate padding), MimiSplitResidualVectorQuantizer.encode (1 semantic + 15 acoustic
 *   nearest-neighbour layers on 256-d projections, codebook = embed_sum / max(cluster_usage, 1e-5)).
 * Pinned by tests/test_oracle_vs_hf.py against HF's MimiModel run in the build container (tests/make_golden_hf.py);
 * parity with the reference BINARY is unpinned (candle cannot be built here). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "q3_oracle.h"

#define MIMI_MAX_T 256
typedef struct { char name[160]; float* data; int64_t n; } mimi_tensor;
struct q3o_mimi {
    q3o_mimi_config cfg;
    mimi_tensor* t; int n_t, cap_t;
    char err[256];
};

q3o_mimi* q3o_mimi_new(const q3o_mimi_config* cfg) {
    q3o_mimi* m = (q3o_mimi*)calloc(1, sizeof *m);
    m->cfg = *cfg;
    return m;
}
void q3o_mimi_free(q3o_mimi* m) {
    if (!m) return;
    for (int i = 0; i < m->n_t; ++i) free(m->t[i].data);
    free(m->t); free(m);
}
const char* q3o_mimi_last_error(const q3o_mimi* m) { return m->err; }
int q3o_mimi_set_tensor(q3o_mimi* m, const char* name, const float* data, int64_t n) {
    for (int i = 0; i < m->n_t; ++i)
        if (!strcmp(m->t[i].name, name)) { free(m->t[i].data); m->t[i].data = NULL; m->t[i] = m->t[--m->n_t]; break; }
    if (m->n_t == m->cap_t) { m->cap_t = m->cap_t ? 2 * m->cap_t : 64; m->t = (mimi_tensor*)realloc(m->t, (size_t)m->cap_t * sizeof *m->t); }
    mimi_tensor* t = &m->t[m->n_t++];
    snprintf(t->name, sizeof t->name, "%s", name);
    t->n = n; t->data = (float*)malloc((size_t)(n ? n : 1) * sizeof(float));
    memcpy(t->data, data, (size_t)n * sizeof(float));
    return 0;
}
static const float* T(q3o_mimi* m, const char* name, int64_t n) {
    for (int i = 0; i < m->n_t; ++i)
        if (!strcmp(m->t[i].name, name)) {
            if (m->t[i].n != n) { snprintf(m->err, sizeof m->err, "tensor %s has %lld elements, expected %lld", name, (long long)m->t[i].n, (long long)n); return NULL; }
            return m->t[i].data;
        }
    snprintf(m->err, sizeof m->err, "Missing weight: %s", name);
    return NULL;
}

static int ceil_div(int a, int b) { return (a + b - 1) / b; }
/* frames at the codec rate for n samples: every MimiConv1d yields ceil(L / stride) outputs (modeling_mimi.py MimiConv1d._get_output_length) */
int q3o_mimi_frames(const q3o_mimi_config* c, int64_t n_samples) {
    int64_t L = n_samples;
    for (int i = 0; i < 4; ++i) L = (L + c->ratios[i] - 1) / c->ratios[i];
    return (int)((L + 1) / 2);
}

/* MimiConv1d.forward, causal: y[co][t] = b[co] + sum_ci sum_kk w[co][ci][kk] * xp[ci][t*stride + kk*dil],
 * xp = x left-padded by (k-1)*dil + 1 - stride and right-padded so the last frame is whole; replicate = edge values */
static float* conv1d_mimi(const float* x, int cin, int L, const float* w, const float* b, int cout, int k, int stride, int dil,
                          int replicate, int* Lout_p) {
    int keff = (k - 1) * dil + 1, pad_total = keff - stride;
    int Lout = ceil_div(L, stride);
    int ideal = (Lout - 1) * stride + keff - pad_total;
    int extra = ideal - L; if (extra < 0) extra = 0;
    int Lp = L + pad_total + extra;
    float* xp = (float*)malloc((size_t)cin * Lp * sizeof(float));
    for (int ci = 0; ci < cin; ++ci) {
        float* r = xp + (size_t)ci * Lp; const float* s = x + (size_t)ci * L;
        for (int i = 0; i < pad_total; ++i) r[i] = replicate ? s[0] : 0.0f;
        memcpy(r + pad_total, s, (size_t)L * sizeof(float));
        for (int i = 0; i < extra; ++i) r[pad_total + L + i] = replicate ? s[L - 1] : 0.0f;
    }
    float* y = (float*)malloc((size_t)cout * Lout * sizeof(float));
#pragma omp parallel for schedule(static) if ((double)cout * cin * k * Lout > 2.0e5)
    for (int co = 0; co < cout; ++co) {
        float* yr = y + (size_t)co * Lout;
        for (int t = 0; t < Lout; ++t) yr[t] = 0.0f;
        for (int ci = 0; ci < cin; ++ci) {
            const float* xr = xp + (size_t)ci * Lp; const float* wr = w + ((size_t)co * cin + ci) * k;
            for (int kk = 0; kk < k; ++kk) {
                float wv = wr[kk]; const float* xs = xr + kk * dil;
                for (int t = 0; t < Lout; ++t) yr[t] += wv * xs[(size_t)t * stride];
            }
        }
        if (b) { float bv = b[co]; for (int t = 0; t < Lout; ++t) yr[t] += bv; }
    }
    free(xp);
    *Lout_p = Lout;
    return y;
}
static void elu_inplace(float* x, size_t n) { for (size_t i = 0; i < n; ++i) if (x[i] <= 0.0f) x[i] = expf(x[i]) - 1.0f; }   /* nn.ELU(alpha = 1) */

#define NAME(buf, ...) snprintf(buf, sizeof buf, __VA_ARGS__)
#define GET(var, n, ...) do { char nm_[160]; NAME(nm_, __VA_ARGS__); var = T(m, nm_, (int64_t)(n)); if (!var) return -1; } while (0)

/* samples → codes [T][n_q]; taps (optional, may be NULL): 0 = SEANet output [hidden][T25], 1 = transformer output [hidden][T25],
 * 2 = downsampled [hidden][T], 3 = per-decision margin [T][n_q] (second-best minus best squared distance). Returns T or -1. */
int q3o_mimi_encode(q3o_mimi* m, const float* samples, int64_t n, uint32_t* codes, float** taps) {
    const q3o_mimi_config* c = &m->cfg;
    if (n < 1) { snprintf(m->err, sizeof m->err, "empty audio"); return -1; }
    const int F = c->n_filters, H = c->hidden;
    /* ---- MimiEncoder (SEANet) ---- */
    const float *w, *b;
    int L = (int)n, Ln;
    GET(w, (int64_t)F * 1 * c->kernel, "encoder.encoder.layers.0.conv.weight"); GET(b, F, "encoder.encoder.layers.0.conv.bias");
    float* x = conv1d_mimi(samples, 1, L, w, b, F, c->kernel, 1, 1, 0, &Ln); L = Ln;
    int dim = F, li = 1;
    for (int s = 0; s < 4; ++s) {
        const int r = c->ratios[s], hid = dim / c->compress;
        /* MimiResnetBlock: x + conv1(ELU(conv3(ELU(x)))) (identity shortcut) */
        float* h = (float*)malloc((size_t)dim * L * sizeof(float)); memcpy(h, x, (size_t)dim * L * sizeof(float));
        elu_inplace(h, (size_t)dim * L);
        GET(w, (int64_t)hid * dim * c->res_kernel, "encoder.encoder.layers.%d.block.1.conv.weight", li); GET(b, hid, "encoder.encoder.layers.%d.block.1.conv.bias", li);
        float* h2 = conv1d_mimi(h, dim, L, w, b, hid, c->res_kernel, 1, 1, 0, &Ln); free(h);
        elu_inplace(h2, (size_t)hid * L);
        GET(w, (int64_t)dim * hid, "encoder.encoder.layers.%d.block.3.conv.weight", li); GET(b, dim, "encoder.encoder.layers.%d.block.3.conv.bias", li);
        float* h3 = conv1d_mimi(h2, hid, L, w, b, dim, 1, 1, 1, 0, &Ln); free(h2);
        for (size_t i = 0; i < (size_t)dim * L; ++i) x[i] = x[i] + h3[i];
        free(h3);
        /* ELU + strided conv (kernel 2r, stride r, channels doubled) */
        elu_inplace(x, (size_t)dim * L);
        GET(w, (int64_t)2 * dim * dim * 2 * r, "encoder.encoder.layers.%d.conv.weight", li + 2); GET(b, 2 * dim, "encoder.encoder.layers.%d.conv.bias", li + 2);
        float* y = conv1d_mimi(x, dim, L, w, b, 2 * dim, 2 * r, r, 1, 0, &Ln); free(x);
        x = y; L = Ln; dim *= 2; li += 3;
    }
    elu_inplace(x, (size_t)dim * L);
    GET(w, (int64_t)H * dim * c->last_kernel, "encoder.encoder.layers.%d.conv.weight", li + 1); GET(b, H, "encoder.encoder.layers.%d.conv.bias", li + 1);
    { float* y = conv1d_mimi(x, dim, L, w, b, H, c->last_kernel, 1, 1, 0, &Ln); free(x); x = y; L = Ln; }
    if (taps && taps[0]) memcpy(taps[0], x, (size_t)H * L * sizeof(float));

    /* ---- MimiTransformerModel: rows [T][H] ---- */
    const int Tn = L, nh = c->n_heads, hd = c->head_dim, QD = nh * hd, I = c->inter, half = hd / 2;
    float* hs = (float*)malloc((size_t)Tn * H * sizeof(float));
    for (int t = 0; t < Tn; ++t) for (int i = 0; i < H; ++i) hs[(size_t)t * H + i] = x[(size_t)i * Tn + t];
    free(x);
    float* cs = (float*)malloc((size_t)Tn * half * sizeof(float)); float* sn = (float*)malloc((size_t)Tn * half * sizeof(float));
    q3o_rope_table(c->rope_theta, hd, 0, Tn, cs, sn);       /* MimiRotaryEmbedding: inv_freq = theta^(-2i/d), freqs = pos * inv_freq (f32) */
    float* nrm = (float*)malloc((size_t)Tn * H * sizeof(float));
    float* q = (float*)malloc((size_t)Tn * QD * sizeof(float)); float* k = (float*)malloc((size_t)Tn * QD * sizeof(float));
    float* v = (float*)malloc((size_t)Tn * QD * sizeof(float)); float* att = (float*)malloc((size_t)Tn * QD * sizeof(float));
    float* ao = (float*)malloc((size_t)Tn * H * sizeof(float)); float* ff = (float*)malloc((size_t)Tn * I * sizeof(float));
    const float scale = 1.0f / sqrtf((float)hd);
    int rc = 0;
    for (int l = 0; l < c->n_layers && !rc; ++l) {
        const float *lnw, *lnb, *wq, *wk, *wv, *wo, *sa, *pw, *pb, *f1, *f2, *sm;
#define GETL(var, n, suffix) do { char nm_[160]; NAME(nm_, "encoder.encoder_transformer.layers.%d.%s", l, suffix); var = T(m, nm_, (int64_t)(n)); if (!var) { rc = -1; } } while (0)
        GETL(lnw, H, "input_layernorm.weight"); GETL(lnb, H, "input_layernorm.bias");
        GETL(wq, (int64_t)QD * H, "self_attn.q_proj.weight"); GETL(wk, (int64_t)QD * H, "self_attn.k_proj.weight");
        GETL(wv, (int64_t)QD * H, "self_attn.v_proj.weight"); GETL(wo, (int64_t)H * QD, "self_attn.o_proj.weight");
        GETL(sa, H, "self_attn_layer_scale.scale"); GETL(pw, H, "post_attention_layernorm.weight"); GETL(pb, H, "post_attention_layernorm.bias");
        GETL(f1, (int64_t)I * H, "mlp.fc1.weight"); GETL(f2, (int64_t)H * I, "mlp.fc2.weight"); GETL(sm, H, "mlp_layer_scale.scale");
        if (rc) break;
        for (int pass = 0; pass < 2; ++pass) {
            const float* gw = pass ? pw : lnw; const float* gb = pass ? pb : lnb;
            /* nn.LayerNorm: (x - mean) / sqrt(var + eps) * w + b, biased variance */
            for (int t = 0; t < Tn; ++t) {
                const float* r = hs + (size_t)t * H; float s1 = 0.0f;
                for (int i = 0; i < H; ++i) s1 += r[i];
                float mean = s1 / (float)H, s2 = 0.0f;
                for (int i = 0; i < H; ++i) { float d = r[i] - mean; s2 += d * d; }
                float inv = 1.0f / sqrtf(s2 / (float)H + c->norm_eps);
                for (int i = 0; i < H; ++i) nrm[(size_t)t * H + i] = (r[i] - mean) * inv * gw[i] + gb[i];
            }
            if (pass == 0) {
                q3o_linear(nrm, wq, NULL, q, Tn, QD, H); q3o_linear(nrm, wk, NULL, k, Tn, QD, H); q3o_linear(nrm, wv, NULL, v, Tn, QD, H);
                /* apply_rotary_pos_emb: x*cos + rotate_half(x)*sin, cos/sin = cat(freqs, freqs) */
                for (int t = 0; t < Tn; ++t)
                    for (int hh = 0; hh < 2 * nh; ++hh) {
                        float* p = (hh < nh ? q : k) + (size_t)t * QD + (size_t)(hh % nh) * hd;
                        float tmp[256];
                        for (int i = 0; i < half; ++i) {
                            float x1 = p[i], x2 = p[i + half], cv = cs[(size_t)t * half + i], sv = sn[(size_t)t * half + i];
                            tmp[i] = x1 * cv + (-x2) * sv; tmp[i + half] = x2 * cv + x1 * sv;
                        }
                        memcpy(p, tmp, (size_t)hd * sizeof(float));
                    }
#pragma omp parallel for collapse(2) schedule(static) if ((double)Tn * Tn * QD > 2.0e5)
                for (int t = 0; t < Tn; ++t)
                    for (int hh = 0; hh < nh; ++hh) {
                        /* causal sliding window: keys j with t - window < j <= t (sliding_window_causal mask of modeling_mimi.py) */
                        int j0 = t - c->window + 1; if (j0 < 0) j0 = 0;
                        float sc[1024]; float* scp = (t - j0 + 1) <= 1024 ? sc : (float*)malloc((size_t)(t - j0 + 1) * sizeof(float));
                        const float* qh = q + (size_t)t * QD + (size_t)hh * hd;
                        float mx = -INFINITY;
                        for (int j = j0; j <= t; ++j) {
                            const float* kh = k + (size_t)j * QD + (size_t)hh * hd; float d = 0.0f;
                            for (int e = 0; e < hd; ++e) d += qh[e] * kh[e];
                            scp[j - j0] = d * scale; if (scp[j - j0] > mx) mx = scp[j - j0];
                        }
                        float sum = 0.0f;
                        for (int j = j0; j <= t; ++j) { scp[j - j0] = expf(scp[j - j0] - mx); sum += scp[j - j0]; }
                        float* o = att + (size_t)t * QD + (size_t)hh * hd;
                        for (int e = 0; e < hd; ++e) o[e] = 0.0f;
                        for (int j = j0; j <= t; ++j) {
                            float pj = scp[j - j0] / sum; const float* vj = v + (size_t)j * QD + (size_t)hh * hd;
                            for (int e = 0; e < hd; ++e) o[e] += pj * vj[e];
                        }
                        if (scp != sc) free(scp);
                    }
                q3o_linear(att, wo, NULL, ao, Tn, H, QD);
                for (size_t i = 0; i < (size_t)Tn * H; ++i) hs[i] = hs[i] + sa[i % H] * ao[i];          /* residual + LayerScale */
            } else {
                q3o_linear(nrm, f1, NULL, ff, Tn, I, H);
                for (size_t i = 0; i < (size_t)Tn * I; ++i) { float z = ff[i]; ff[i] = 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); }   /* ACT2FN["gelu"] */
                q3o_linear(ff, f2, NULL, ao, Tn, H, I);
                for (size_t i = 0; i < (size_t)Tn * H; ++i) hs[i] = hs[i] + sm[i % H] * ao[i];
            }
        }
    }
    free(cs); free(sn); free(nrm); free(q); free(k); free(v); free(att); free(ao); free(ff);
    if (rc) { free(hs); return -1; }
    float* xt = (float*)malloc((size_t)H * Tn * sizeof(float));
    for (int t = 0; t < Tn; ++t) for (int i = 0; i < H; ++i) xt[(size_t)i * Tn + t] = hs[(size_t)t * H + i];
    free(hs);
    if (taps && taps[1]) memcpy(taps[1], xt, (size_t)H * Tn * sizeof(float));

    /* ---- downsample: MimiConv1d(k = 4, stride 2, no bias, pad_mode "replicate") ---- */
    GET(w, (int64_t)H * H * 4, "encoder.downsample.conv.weight");
    int T12;
    float* xd = conv1d_mimi(xt, H, Tn, w, NULL, H, 4, 2, 1, 1, &T12); free(xt);
    if (taps && taps[2]) memcpy(taps[2], xd, (size_t)H * T12 * sizeof(float));

    /* ---- MimiSplitResidualVectorQuantizer.encode ---- */
    const int CD = c->cb_dim, CB = c->cb_size;
    for (int grp = 0; grp < 2; ++grp) {
        const char* gname = grp ? "acoustic_residual_vector_quantizer" : "semantic_residual_vector_quantizer";
        const int nl = grp ? c->n_q - c->n_sem : c->n_sem, q0 = grp ? c->n_sem : 0;
        GET(w, (int64_t)CD * H, "encoder.quantizer.%s.input_proj.weight", gname);
        int Lq; float* p = conv1d_mimi(xd, H, T12, w, NULL, CD, 1, 1, 1, 0, &Lq);        /* [CD][T] */
        float* res = (float*)malloc((size_t)T12 * CD * sizeof(float));
        for (int t = 0; t < T12; ++t) for (int d = 0; d < CD; ++d) res[(size_t)t * CD + d] = p[(size_t)d * T12 + t];
        free(p);
        float* emb = (float*)malloc((size_t)CB * CD * sizeof(float));
        for (int l = 0; l < nl; ++l) {
            const float *es, *cu;
            GET(es, (int64_t)CB * CD, "encoder.quantizer.%s.layers.%d.codebook.embed_sum", gname, l);
            GET(cu, CB, "encoder.quantizer.%s.layers.%d.codebook.cluster_usage", gname, l);
            for (int e = 0; e < CB; ++e) { float u = cu[e] < 1e-5f ? 1e-5f : cu[e]; for (int d = 0; d < CD; ++d) emb[(size_t)e * CD + d] = es[(size_t)e * CD + d] / u; }
#pragma omp parallel for schedule(static) if ((double)T12 * CB * CD > 2.0e5)
            for (int t = 0; t < T12; ++t) {
                float* r = res + (size_t)t * CD; int best = 0; float bd = INFINITY, second = INFINITY;
                for (int e = 0; e < CB; ++e) {                        /* torch.cdist(p = 2).argmin: first minimum */
                    const float* ev = emb + (size_t)e * CD; float d2 = 0.0f;
                    for (int d = 0; d < CD; ++d) { float df = r[d] - ev[d]; d2 += df * df; }
                    if (d2 < bd) { second = bd; bd = d2; best = e; } else if (d2 < second) second = d2;
                }
                codes[(size_t)t * c->n_q + q0 + l] = (uint32_t)best;
                if (taps && taps[3]) taps[3][(size_t)t * c->n_q + q0 + l] = second - bd;     /* squared-distance margin of the decision */
                const float* ev = emb + (size_t)best * CD;
                for (int d = 0; d < CD; ++d) r[d] = r[d] - ev[d];
            }
        }
        free(emb); free(res);
    }
    free(xd);
    return T12;
}
