/*
 * q3_oracle.c — CPU F32 restatement of the reference hot path (see q3_oracle.h header for the
 * "TEST INFRASTRUCTURE ONLY" / "parity unpinned" statements).
 *
 * Numeric conventions: everything is IEEE f32, compiled with
 * -ffp-contract=off so no FMA is formed behind our back. candle's gemm summation order is not
 * knowable here; dot products use a FIXED order (16 strided partial sums combined by a
 * pairwise tree, `dot_f32`) so results do not depend on thread count. Rows of a matmul are
 * distributed over OpenMP threads; each output element is computed by exactly one thread.
 */
#include "q3_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- token ids: talker.rs:30-55 ---- */
enum { IM_START = 151644, ASSISTANT = 77091, NEWLINE = 198 };
enum { TTS_PAD = 151671, TTS_BOS = 151672, TTS_EOS = 151673 };
enum { CODEC_PAD = 2148, CODEC_BOS = 2149, CODEC_EOS = 2150, CODEC_THINK = 2154,
       CODEC_NOTHINK = 2155, CODEC_THINK_BOS = 2156, CODEC_THINK_EOS = 2157 };

static __thread char g_err[512];
const char* q3o_last_error(void) { return g_err; }
static int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
/* Thread policy: parallel regions are used only when there is enough work (PAR_IF), and never with
 * more than g_threads threads (default min(cores, 32)): on a 256-thread host a fork/join per tiny
 * matmul costs far more than the matmul. Results do not depend on the thread count. */
static int g_threads = 0;
static int n_threads(void) {
    if (g_threads > 0) return g_threads;
#ifdef _OPENMP
    int n = omp_get_num_procs();
    g_threads = n > 32 ? 32 : (n < 1 ? 1 : n);
#else
    g_threads = 1;
#endif
    return g_threads;
}
void q3o_set_threads(int n) { if (n > 0) g_threads = n; }
int q3o_get_threads(void) { return n_threads(); }
#define PAR_IF(work) if ((double)(work) > 2.0e5) num_threads(n_threads())

static float* fmalloc(size_t n) {
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    if (!p) { fprintf(stderr, "q3_oracle: out of memory (%zu floats)\n", n); abort(); }
    return p;
}
static float* fcalloc(size_t n) {
    float* p = (float*)calloc(n ? n : 1, sizeof(float));
    if (!p) { fprintf(stderr, "q3_oracle: out of memory (%zu floats)\n", n); abort(); }
    return p;
}

/* ------------------------------------------------------------------------------------------
 * basic math
 * ---------------------------------------------------------------------------------------- */
static inline float dot_f32(const float* a, const float* b, int n) {
    float acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    int i = 0;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; ++j) acc[j] += a[i + j] * b[i + j];
    for (int j = 0; i < n; ++i, ++j) acc[j] += a[i] * b[i];
    for (int s = 8; s >= 1; s >>= 1)
        for (int j = 0; j < s; ++j) acc[j] += acc[j + s];
    return acc[0];
}

/* Linear: y = x·Wᵀ (+b), W [N][K] row-major (candle_nn::Linear) */
void q3o_linear(const float* x, const float* w, const float* b, float* y, int M, int N, int K) {
#pragma omp parallel for schedule(static) PAR_IF((double)M * N * K)
    for (int n = 0; n < N; ++n) {
        const float* wr = w + (size_t)n * K;
        for (int m = 0; m < M; ++m) {
            float v = dot_f32(x + (size_t)m * K, wr, K);
            if (b) v = v + b[n];
            y[(size_t)m * N + n] = v;
        }
    }
}

/* candle rms_norm CPU op: m = sqrt(sum(x²)/n + eps); y = x / m * w  (Appendix A.2) */
void q3o_rms_norm(const float* x, const float* w, float* y, int rows, int cols, float eps) {
    for (int r = 0; r < rows; ++r) {
        const float* xr = x + (size_t)r * cols;
        float* yr = y + (size_t)r * cols;
        float ss = 0.0f;
        for (int c = 0; c < cols; ++c) ss += xr[c] * xr[c];
        float m = sqrtf(ss / (float)cols + eps);
        for (int c = 0; c < cols; ++c) yr[c] = xr[c] / m * w[c];
    }
}

/* fused_ops.rs:59-67 (CPU branch): sum = x + residual; normed = rms_norm(sum) */
void q3o_fused_residual_rmsnorm(const float* x, const float* res, const float* w, int rows, int cols,
                                float eps, float* normed, float* sum) {
    size_t n = (size_t)rows * cols;
    for (size_t i = 0; i < n; ++i) sum[i] = x[i] + res[i];
    q3o_rms_norm(sum, w, normed, rows, cols, eps);
}

/* transformer.rs:78-92, 133-175: inv_freq in f32 powf; freqs = pos(f32) * inv_freq */
void q3o_rope_table(float theta, int head_dim, int pos0, int n_pos, float* cos_out, float* sin_out) {
    int half = head_dim / 2;
    for (int p = 0; p < n_pos; ++p)
        for (int i = 0; i < half; ++i) {
            float inv = 1.0f / powf(theta, (float)(2 * i) / (float)head_dim);
            float f = (float)(pos0 + p) * inv;
            cos_out[(size_t)p * half + i] = cosf(f);
            sin_out[(size_t)p * half + i] = sinf(f);
        }
}

static inline float silu_f(float v) { return v / (1.0f + expf(-v)); }

/* ------------------------------------------------------------------------------------------
 * model container
 * ---------------------------------------------------------------------------------------- */
typedef struct { char* name; float* data; int64_t n; } tensor_t;

typedef struct {
    const float *in_ln, *q, *k, *v, *o, *q_norm, *k_norm, *post_ln, *gate, *up, *down;
} layer_w;

typedef struct {
    const float *in_ln, *q, *k, *v, *o, *attn_scale, *post_ln, *gate, *up, *down, *mlp_scale;
} dec_layer_w;

typedef struct {
    const float *a1, *b1, *c1w, *c1b, *a2, *b2, *c2w, *c2b;
} res_unit_w;

typedef struct {
    const float *alpha, *beta, *tw, *tb;
    res_unit_w res[3];
} dec_block_w;

typedef struct {
    const float *tw, *tb, *dww, *dwb, *nw, *nb, *p1w, *p1b, *p2w, *p2b, *gamma;
} upsample_w;

struct q3o_model {
    q3o_config cfg;
    tensor_t* t; int nt, cap;
    /* talker */
    const float *text_emb, *fc1w, *fc1b, *fc2w, *fc2b, *codec_emb, *norm, *codec_head;
    layer_w* tl;
    /* code predictor */
    const float *mtp_w, *mtp_b, *cp_norm;
    const float* cp_emb[32]; const float* cp_head[32];
    layer_w* cl;
    int cp_embed_dim;
    /* decoder */
    float* first_cb; float* rest_cb[15];
    const float *first_proj, *rest_proj, *pre_w, *pre_b, *inp_w, *inp_b, *outp_w, *outp_b, *dec_norm;
    dec_layer_w* dl;
    upsample_w up[2];
    const float *init_w, *init_b;
    dec_block_w blk[4];
    const float *fin_a, *fin_b, *fin_w, *fin_bias;
    int have_lm, have_dec;
};

q3o_model* q3o_model_new(const q3o_config* cfg) {
    q3o_model* m = (q3o_model*)calloc(1, sizeof *m);
    m->cfg = *cfg;
    return m;
}

void q3o_model_free(q3o_model* m) {
    if (!m) return;
    for (int i = 0; i < m->nt; ++i) { free(m->t[i].name); free(m->t[i].data); }
    free(m->t); free(m->tl); free(m->cl); free(m->dl);
    free(m->first_cb);
    for (int i = 0; i < 15; ++i) free(m->rest_cb[i]);
    free(m);
}

int q3o_model_set_tensor(q3o_model* m, const char* name, const float* data, int64_t n) {
    if (m->nt == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 256;
        m->t = (tensor_t*)realloc(m->t, (size_t)m->cap * sizeof(tensor_t));
    }
    tensor_t* t = &m->t[m->nt++];
    t->name = strdup(name);
    t->data = fmalloc((size_t)n);
    memcpy(t->data, data, (size_t)n * sizeof(float));
    t->n = n;
    return 0;
}

static const float* find(q3o_model* m, int* bad, int64_t expect, const char* fmt, ...) {
    char name[256];
    va_list ap; va_start(ap, fmt); vsnprintf(name, sizeof name, fmt, ap); va_end(ap);
    for (int i = 0; i < m->nt; ++i)
        if (strcmp(m->t[i].name, name) == 0) {
            if (expect >= 0 && m->t[i].n != expect) {
                if (!*bad) fail("tensor %s has %lld elements, expected %lld", name, (long long)m->t[i].n, (long long)expect);
                *bad = 1;
            }
            return m->t[i].data;
        }
    if (!*bad) fail("Missing weight: %s", name);   /* lib.rs:226-231, decoder_12hz.rs:176-181 */
    *bad = 1;
    return NULL;
}

static void load_layer(q3o_model* m, int* bad, layer_w* L, const char* prefix, int i, int H, int I, int nh, int nkv, int hd) {
    L->in_ln   = find(m, bad, H, "%s.layers.%d.input_layernorm.weight", prefix, i);
    L->q       = find(m, bad, (int64_t)nh * hd * H, "%s.layers.%d.self_attn.q_proj.weight", prefix, i);
    L->k       = find(m, bad, (int64_t)nkv * hd * H, "%s.layers.%d.self_attn.k_proj.weight", prefix, i);
    L->v       = find(m, bad, (int64_t)nkv * hd * H, "%s.layers.%d.self_attn.v_proj.weight", prefix, i);
    L->o       = find(m, bad, (int64_t)H * nh * hd, "%s.layers.%d.self_attn.o_proj.weight", prefix, i);
    L->q_norm  = find(m, bad, hd, "%s.layers.%d.self_attn.q_norm.weight", prefix, i);
    L->k_norm  = find(m, bad, hd, "%s.layers.%d.self_attn.k_norm.weight", prefix, i);
    L->post_ln = find(m, bad, H, "%s.layers.%d.post_attention_layernorm.weight", prefix, i);
    L->gate    = find(m, bad, (int64_t)I * H, "%s.layers.%d.mlp.gate_proj.weight", prefix, i);
    L->up      = find(m, bad, (int64_t)I * H, "%s.layers.%d.mlp.up_proj.weight", prefix, i);
    L->down    = find(m, bad, (int64_t)H * I, "%s.layers.%d.mlp.down_proj.weight", prefix, i);
}

/* decoder_12hz.rs:189-225: codebook = embedding_sum / clamp(cluster_usage, 1e-7, f32::MAX) */
static float* norm_codebook(const float* esum, const float* usage, int rows, int dim) {
    float* cb = fmalloc((size_t)rows * dim);
    for (int r = 0; r < rows; ++r) {
        float u = usage[r];
        if (u < 1e-7f) u = 1e-7f;
        for (int d = 0; d < dim; ++d) cb[(size_t)r * dim + d] = esum[(size_t)r * dim + d] / u;
    }
    return cb;
}

int q3o_model_finalize(q3o_model* m, int which) {
    const q3o_config* c = &m->cfg;
    int bad = 0;
    if (which & 1) {
        int H = c->hidden, TD = c->text_dim;
        m->text_emb   = find(m, &bad, (int64_t)c->text_vocab * TD, "talker.model.text_embedding.weight");
        m->fc1w       = find(m, &bad, (int64_t)TD * TD, "talker.text_projection.linear_fc1.weight");
        m->fc1b       = find(m, &bad, TD, "talker.text_projection.linear_fc1.bias");
        m->fc2w       = find(m, &bad, (int64_t)H * TD, "talker.text_projection.linear_fc2.weight");
        m->fc2b       = find(m, &bad, H, "talker.text_projection.linear_fc2.bias");
        m->codec_emb  = find(m, &bad, (int64_t)c->codec_vocab * H, "talker.model.codec_embedding.weight");
        m->norm       = find(m, &bad, H, "talker.model.norm.weight");
        m->codec_head = find(m, &bad, (int64_t)c->codec_vocab * H, "talker.codec_head.weight");
        m->tl = (layer_w*)calloc((size_t)c->n_layers, sizeof(layer_w));
        for (int i = 0; i < c->n_layers; ++i)
            load_layer(m, &bad, &m->tl[i], "talker.model", i, H, c->inter, c->n_heads, c->n_kv_heads, c->head_dim);
        /* code predictor (code_predictor.rs:158-234); codec_embed_dim = talker hidden */
        int CH = c->cp_hidden;
        m->cp_embed_dim = H;
        if (H != CH) {
            m->mtp_w = find(m, &bad, (int64_t)CH * H, "talker.code_predictor.small_to_mtp_projection.weight");
            m->mtp_b = find(m, &bad, CH, "talker.code_predictor.small_to_mtp_projection.bias");
        }
        for (int g = 0; g < c->n_groups - 1; ++g) {
            m->cp_emb[g]  = find(m, &bad, (int64_t)c->cp_vocab * H, "talker.code_predictor.model.codec_embedding.%d.weight", g);
            m->cp_head[g] = find(m, &bad, (int64_t)c->cp_vocab * CH, "talker.code_predictor.lm_head.%d.weight", g);
        }
        m->cp_norm = find(m, &bad, CH, "talker.code_predictor.model.norm.weight");
        m->cl = (layer_w*)calloc((size_t)c->cp_layers, sizeof(layer_w));
        for (int i = 0; i < c->cp_layers; ++i)
            load_layer(m, &bad, &m->cl[i], "talker.code_predictor.model", i, CH, c->cp_inter, c->cp_heads, c->cp_kv_heads, c->head_dim);
        if (!bad) m->have_lm = 1;
    }
    if (which & 2) {
        int CB = c->dec_cb_size, CD = c->dec_cb_dim, Q = c->dec_q_dim, LAT = c->dec_latent, DH = c->dec_hidden;
        const float* es = find(m, &bad, (int64_t)CB * CD, "decoder.quantizer.rvq_first.vq.layers.0._codebook.embedding_sum");
        const float* us = find(m, &bad, CB, "decoder.quantizer.rvq_first.vq.layers.0._codebook.cluster_usage");
        if (es && us) m->first_cb = norm_codebook(es, us, CB, CD);
        for (int i = 0; i < 15; ++i) {
            es = find(m, &bad, (int64_t)CB * CD, "decoder.quantizer.rvq_rest.vq.layers.%d._codebook.embedding_sum", i);
            us = find(m, &bad, CB, "decoder.quantizer.rvq_rest.vq.layers.%d._codebook.cluster_usage", i);
            if (es && us) m->rest_cb[i] = norm_codebook(es, us, CB, CD);
        }
        m->first_proj = find(m, &bad, (int64_t)Q * CD, "decoder.quantizer.rvq_first.output_proj.weight");
        m->rest_proj  = find(m, &bad, (int64_t)Q * CD, "decoder.quantizer.rvq_rest.output_proj.weight");
        m->pre_w = find(m, &bad, (int64_t)LAT * Q * 3, "decoder.pre_conv.conv.weight");
        m->pre_b = find(m, &bad, LAT, "decoder.pre_conv.conv.bias");
        m->inp_w = find(m, &bad, (int64_t)DH * LAT, "decoder.pre_transformer.input_proj.weight");
        m->inp_b = find(m, &bad, DH, "decoder.pre_transformer.input_proj.bias");
        m->outp_w = find(m, &bad, (int64_t)LAT * DH, "decoder.pre_transformer.output_proj.weight");
        m->outp_b = find(m, &bad, LAT, "decoder.pre_transformer.output_proj.bias");
        m->dec_norm = find(m, &bad, DH, "decoder.pre_transformer.norm.weight");
        int QD = c->dec_heads * c->dec_head_dim, DI = c->dec_inter;
        m->dl = (dec_layer_w*)calloc((size_t)c->dec_layers, sizeof(dec_layer_w));
        for (int i = 0; i < c->dec_layers; ++i) {
            dec_layer_w* L = &m->dl[i];
            const char* p = "decoder.pre_transformer.layers";
            L->in_ln = find(m, &bad, DH, "%s.%d.input_layernorm.weight", p, i);
            L->q = find(m, &bad, (int64_t)QD * DH, "%s.%d.self_attn.q_proj.weight", p, i);
            L->k = find(m, &bad, (int64_t)QD * DH, "%s.%d.self_attn.k_proj.weight", p, i);
            L->v = find(m, &bad, (int64_t)QD * DH, "%s.%d.self_attn.v_proj.weight", p, i);
            L->o = find(m, &bad, (int64_t)DH * QD, "%s.%d.self_attn.o_proj.weight", p, i);
            L->attn_scale = find(m, &bad, DH, "%s.%d.self_attn_layer_scale.scale", p, i);
            L->post_ln = find(m, &bad, DH, "%s.%d.post_attention_layernorm.weight", p, i);
            L->gate = find(m, &bad, (int64_t)DI * DH, "%s.%d.mlp.gate_proj.weight", p, i);
            L->up = find(m, &bad, (int64_t)DI * DH, "%s.%d.mlp.up_proj.weight", p, i);
            L->down = find(m, &bad, (int64_t)DH * DI, "%s.%d.mlp.down_proj.weight", p, i);
            L->mlp_scale = find(m, &bad, DH, "%s.%d.mlp_layer_scale.scale", p, i);
        }
        for (int i = 0; i < 2; ++i) {
            upsample_w* U = &m->up[i];
            int r = c->dec_up_ratios[i];
            U->tw = find(m, &bad, (int64_t)LAT * LAT * r, "decoder.upsample.%d.0.conv.weight", i);
            U->tb = find(m, &bad, LAT, "decoder.upsample.%d.0.conv.bias", i);
            U->dww = find(m, &bad, (int64_t)LAT * 7, "decoder.upsample.%d.1.dwconv.conv.weight", i);
            U->dwb = find(m, &bad, LAT, "decoder.upsample.%d.1.dwconv.conv.bias", i);
            U->nw = find(m, &bad, LAT, "decoder.upsample.%d.1.norm.weight", i);
            U->nb = find(m, &bad, LAT, "decoder.upsample.%d.1.norm.bias", i);
            U->p1w = find(m, &bad, (int64_t)4 * LAT * LAT, "decoder.upsample.%d.1.pwconv1.weight", i);
            U->p1b = find(m, &bad, 4 * LAT, "decoder.upsample.%d.1.pwconv1.bias", i);
            U->p2w = find(m, &bad, (int64_t)4 * LAT * LAT, "decoder.upsample.%d.1.pwconv2.weight", i);
            U->p2b = find(m, &bad, LAT, "decoder.upsample.%d.1.pwconv2.bias", i);
            U->gamma = find(m, &bad, LAT, "decoder.upsample.%d.1.gamma", i);
        }
        int D = c->dec_dim;
        m->init_w = find(m, &bad, (int64_t)D * LAT * 7, "decoder.decoder.0.conv.weight");
        m->init_b = find(m, &bad, D, "decoder.decoder.0.conv.bias");
        int cin = D;
        for (int b = 0; b < 4; ++b) {
            dec_block_w* B = &m->blk[b];
            int r = c->dec_up_rates[b], cout = cin / 2;
            B->alpha = find(m, &bad, cin, "decoder.decoder.%d.block.0.alpha", b + 1);
            B->beta  = find(m, &bad, cin, "decoder.decoder.%d.block.0.beta", b + 1);
            B->tw = find(m, &bad, (int64_t)cin * cout * 2 * r, "decoder.decoder.%d.block.1.conv.weight", b + 1);
            B->tb = find(m, &bad, cout, "decoder.decoder.%d.block.1.conv.bias", b + 1);
            for (int u = 0; u < 3; ++u) {
                res_unit_w* R = &B->res[u];
                R->a1 = find(m, &bad, cout, "decoder.decoder.%d.block.%d.act1.alpha", b + 1, u + 2);
                R->b1 = find(m, &bad, cout, "decoder.decoder.%d.block.%d.act1.beta", b + 1, u + 2);
                R->c1w = find(m, &bad, (int64_t)cout * cout * 7, "decoder.decoder.%d.block.%d.conv1.conv.weight", b + 1, u + 2);
                R->c1b = find(m, &bad, cout, "decoder.decoder.%d.block.%d.conv1.conv.bias", b + 1, u + 2);
                R->a2 = find(m, &bad, cout, "decoder.decoder.%d.block.%d.act2.alpha", b + 1, u + 2);
                R->b2 = find(m, &bad, cout, "decoder.decoder.%d.block.%d.act2.beta", b + 1, u + 2);
                R->c2w = find(m, &bad, (int64_t)cout * cout, "decoder.decoder.%d.block.%d.conv2.conv.weight", b + 1, u + 2);
                R->c2b = find(m, &bad, cout, "decoder.decoder.%d.block.%d.conv2.conv.bias", b + 1, u + 2);
            }
            cin = cout;
        }
        m->fin_a = find(m, &bad, cin, "decoder.decoder.5.alpha");
        m->fin_b = find(m, &bad, cin, "decoder.decoder.5.beta");
        m->fin_w = find(m, &bad, (int64_t)cin * 7, "decoder.decoder.6.conv.weight");
        m->fin_bias = find(m, &bad, 1, "decoder.decoder.6.conv.bias");
        if (!bad) m->have_dec = 1;
    }
    return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * KV cache (kv_cache.rs:18-91 concat semantics; capacity grows on demand)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float *k, *v; int len, cap, nkv, hd; } kvc_t;

static void kvc_init(kvc_t* c, int nkv, int hd, int cap) {
    c->nkv = nkv; c->hd = hd; c->len = 0; c->cap = cap > 0 ? cap : 32;
    c->k = fmalloc((size_t)nkv * c->cap * hd);
    c->v = fmalloc((size_t)nkv * c->cap * hd);
}
static void kvc_free(kvc_t* c) { free(c->k); free(c->v); c->k = c->v = NULL; }
static void kvc_reserve(kvc_t* c, int need) {
    if (need <= c->cap) return;
    int ncap = c->cap; while (ncap < need) ncap *= 2;
    float* nk = fmalloc((size_t)c->nkv * ncap * c->hd);
    float* nv = fmalloc((size_t)c->nkv * ncap * c->hd);
    for (int h = 0; h < c->nkv; ++h) {
        memcpy(nk + (size_t)h * ncap * c->hd, c->k + (size_t)h * c->cap * c->hd, (size_t)c->len * c->hd * sizeof(float));
        memcpy(nv + (size_t)h * ncap * c->hd, c->v + (size_t)h * c->cap * c->hd, (size_t)c->len * c->hd * sizeof(float));
    }
    free(c->k); free(c->v); c->k = nk; c->v = nv; c->cap = ncap;
}

/* ------------------------------------------------------------------------------------------
 * DecoderLayer::forward (transformer.rs:442-467) with Attention::forward CPU branch
 * (transformer.rs:247-284, 347-371) and MLP::forward (408-413). x: [S][H] in place.
 * `offset` = tokens already cached; S>1 uses the causal mask of transformer.rs:21-36
 * (token i may attend cache positions j <= offset+i); S==1 attends everything.
 * ---------------------------------------------------------------------------------------- */
static void layer_forward(const layer_w* L, float* x, int S, int H, int I, int nh, int nkv, int hd,
                          float eps, float theta, kvc_t* kv, int offset) {
    int QD = nh * hd, KD = nkv * hd, half = hd / 2, n_rep = nh / nkv;
    float* h1 = fmalloc((size_t)S * H);
    float* q = fmalloc((size_t)S * QD);
    float* k = fmalloc((size_t)S * KD);
    float* v = fmalloc((size_t)S * KD);
    float* att = fmalloc((size_t)S * QD);
    float* cs = fmalloc((size_t)S * half);
    float* sn = fmalloc((size_t)S * half);
    float tmp[512];

    q3o_rms_norm(x, L->in_ln, h1, S, H, eps);
    q3o_linear(h1, L->q, NULL, q, S, QD, H);
    q3o_linear(h1, L->k, NULL, k, S, KD, H);
    q3o_linear(h1, L->v, NULL, v, S, KD, H);
    /* per-head q/k RMSNorm before RoPE (transformer.rs:263-269) */
    q3o_rms_norm(q, L->q_norm, q, S * nh, hd, eps);
    q3o_rms_norm(k, L->k_norm, k, S * nkv, hd, eps);
    /* rotate-half RoPE (transformer.rs:42-69) */
    q3o_rope_table(theta, hd, offset, S, cs, sn);
    for (int s = 0; s < S; ++s) {
        for (int hh = 0; hh < nh + nkv; ++hh) {
            float* p = hh < nh ? q + (size_t)s * QD + (size_t)hh * hd : k + (size_t)s * KD + (size_t)(hh - nh) * hd;
            for (int i = 0; i < half; ++i) {
                float x1 = p[i], x2 = p[i + half], c = cs[(size_t)s * half + i], sv = sn[(size_t)s * half + i];
                float a = x1 * c, b = x2 * sv, d = x2 * c, e = x1 * sv;
                tmp[i] = a - b; tmp[i + half] = d + e;
            }
            memcpy(p, tmp, (size_t)hd * sizeof(float));
        }
    }
    /* cache append (kv_cache.rs:290-310) */
    kvc_reserve(kv, offset + S);
    for (int s = 0; s < S; ++s)
        for (int hh = 0; hh < nkv; ++hh) {
            memcpy(kv->k + ((size_t)hh * kv->cap + offset + s) * hd, k + (size_t)s * KD + (size_t)hh * hd, (size_t)hd * sizeof(float));
            memcpy(kv->v + ((size_t)hh * kv->cap + offset + s) * hd, v + (size_t)s * KD + (size_t)hh * hd, (size_t)hd * sizeof(float));
        }
    kv->len = offset + S;
    /* SDPA: repeat_kv (head h uses kv h / n_rep), scores * scale, +mask, softmax, ·V */
    float scale = (float)(1.0 / sqrt((double)hd));
#pragma omp parallel for collapse(2) schedule(static) PAR_IF((double)S * nh * hd * (offset + S) * 4)
    for (int s = 0; s < S; ++s)
        for (int hh = 0; hh < nh; ++hh) {
            int n_ctx = offset + s + 1;      /* S==1: all of the cache; S>1: causal mask */
            const float* qh = q + (size_t)s * QD + (size_t)hh * hd;
            const float* kc = kv->k + (size_t)(hh / n_rep) * kv->cap * hd;
            const float* vc = kv->v + (size_t)(hh / n_rep) * kv->cap * hd;
            float* sc = fmalloc((size_t)n_ctx);
            float mx = -INFINITY;
            for (int j = 0; j < n_ctx; ++j) {
                sc[j] = dot_f32(qh, kc + (size_t)j * hd, hd) * scale;
                if (sc[j] > mx) mx = sc[j];
            }
            float sum = 0.0f;
            for (int j = 0; j < n_ctx; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
            for (int j = 0; j < n_ctx; ++j) sc[j] /= sum;
            float* o = att + (size_t)s * QD + (size_t)hh * hd;
            for (int d = 0; d < hd; ++d) o[d] = 0.0f;
            for (int j = 0; j < n_ctx; ++j) {
                float pj = sc[j]; const float* vj = vc + (size_t)j * hd;
                for (int d = 0; d < hd; ++d) o[d] += pj * vj[d];
            }
            free(sc);
        }
    /* o_proj, fused residual+norm (CPU: sequential), MLP, residual */
    float* ao = fmalloc((size_t)S * H);
    float* sum = fmalloc((size_t)S * H);
    float* nrm = fmalloc((size_t)S * H);
    float* g = fmalloc((size_t)S * I);
    float* u = fmalloc((size_t)S * I);
    q3o_linear(att, L->o, NULL, ao, S, H, QD);
    q3o_fused_residual_rmsnorm(ao, x, L->post_ln, S, H, eps, nrm, sum);
    q3o_linear(nrm, L->gate, NULL, g, S, I, H);
    q3o_linear(nrm, L->up, NULL, u, S, I, H);
    for (size_t i = 0; i < (size_t)S * I; ++i) g[i] = silu_f(g[i]) * u[i];
    q3o_linear(g, L->down, NULL, ao, S, H, I);
    for (size_t i = 0; i < (size_t)S * H; ++i) x[i] = sum[i] + ao[i];
    free(h1); free(q); free(k); free(v); free(att); free(cs); free(sn);
    free(ao); free(sum); free(nrm); free(g); free(u);
}

/* ------------------------------------------------------------------------------------------
 * sampling (generation/sampling.rs) and penalties (lib.rs:1271-1322, generation/tts.rs)
 * ---------------------------------------------------------------------------------------- */
/* sampling.rs:32-51 */
void q3o_rng_seed(uint64_t seed, uint64_t* state) {
    *state = seed * 2685821657736338717ULL + 1442695040888963407ULL;
}
/* sampling.rs:84-94: PCG-XSH-RR 64/32, u = out as f32 / u32::MAX as f32 */
float q3o_rng_next(uint64_t* state) {
    uint64_t old = *state;
    *state = old * 6364136223846793005ULL + 1442695040888963407ULL;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    uint32_t out = (xs >> rot) | (xs << ((32 - rot) & 31));
    return (float)out / (float)UINT32_MAX;
}

/* generation/tts.rs:21-43: suppress [vocab-1024, vocab) except EOS */
void q3o_build_suppression_mask(int vocab, int eos_id, uint8_t* mask) {
    memset(mask, 0, (size_t)vocab);
    for (int v = vocab - 1024; v < vocab; ++v)
        if (v >= 0 && v != eos_id) mask[v] = 1;
}

/* lib.rs:1271-1322: repetition penalty (mask form, sampling.rs:375-400) → suppression
 * (CODEC_EOS hard-wired, lib.rs:543-547) → min_new_tokens EOS mask. eos_id = -1: None. */
void q3o_apply_penalties(float* logits, int vocab, const uint8_t* seen, double rep_penalty,
                         int token_count, int min_new_tokens, int eos_id) {
    if (rep_penalty != 1.0 && !(fabs(rep_penalty - 1.0) < 1e-9)) {
        float pf = (float)rep_penalty, inv = 1.0f / pf;
        for (int i = 0; i < vocab; ++i)
            if (seen[i]) logits[i] = logits[i] * (logits[i] > 0.0f ? inv : pf);
    }
    for (int v = vocab - 1024; v < vocab; ++v)
        if (v >= 0 && v != CODEC_EOS) logits[v] = -INFINITY;
    if (token_count < min_new_tokens && eos_id >= 0 && eos_id < vocab) logits[eos_id] = -INFINITY;
}

typedef struct { float v; int i; } vi_t;
static int cmp_desc(const void* a, const void* b) {
    const vi_t *x = (const vi_t*)a, *y = (const vi_t*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return x->i - y->i;          /* reference sort is unstable; ties broken by index here */
}

/* sampling.rs:189-202 (CPU form): thr = k-th largest; keep v >= thr */
void q3o_top_k_filter(float* logits, int vocab, int k) {
    if (k > vocab) k = vocab;
    if (k <= 0) return;
    vi_t* s = (vi_t*)malloc((size_t)vocab * sizeof(vi_t));
    for (int i = 0; i < vocab; ++i) { s[i].v = logits[i]; s[i].i = i; }
    qsort(s, (size_t)vocab, sizeof(vi_t), cmp_desc);
    float thr = s[k - 1].v;
    for (int i = 0; i < vocab; ++i) if (!(logits[i] >= thr)) logits[i] = -INFINITY;
    free(s);
}

/* sampling.rs:221-262 (CPU form) */
void q3o_top_p_filter(float* logits, int vocab, double p) {
    vi_t* s = (vi_t*)malloc((size_t)vocab * sizeof(vi_t));
    float* e = fmalloc((size_t)vocab);
    for (int i = 0; i < vocab; ++i) { s[i].v = logits[i]; s[i].i = i; }
    qsort(s, (size_t)vocab, sizeof(vi_t), cmp_desc);
    float mx = s[0].v, sum = 0.0f;
    for (int i = 0; i < vocab; ++i) { e[i] = expf(s[i].v - mx); sum += e[i]; }
    for (int i = 0; i < vocab; ++i) e[i] /= sum;
    float cum = 0.0f, pf = (float)p; int cut = vocab;
    for (int i = 0; i < vocab; ++i) { cum += e[i]; if (cum > pf) { cut = i + 1; break; } }
    float* out = fmalloc((size_t)vocab);
    for (int i = 0; i < vocab; ++i) out[i] = -INFINITY;
    for (int i = 0; i < cut; ++i) out[s[i].i] = logits[s[i].i];
    memcpy(logits, out, (size_t)vocab * sizeof(float));
    free(s); free(e); free(out);
}

static uint32_t argmax_first(const float* v, int n) {
    int best = 0;
    for (int i = 1; i < n; ++i) if (v[i] > v[best]) best = i;
    return (uint32_t)best;
}

/* sampling.rs:140-178 + multinomial 290-319 */
uint32_t q3o_sample(const float* logits_in, int vocab, double temperature, int top_k, double top_p,
                    uint64_t* rng_state) {
    float* l = fmalloc((size_t)vocab);
    memcpy(l, logits_in, (size_t)vocab * sizeof(float));
    if (temperature != 1.0 && temperature > 0.0) {
        float mul = (float)(1.0 / temperature);          /* candle Tensor / f64 = affine(1/t, 0) */
        for (int i = 0; i < vocab; ++i) l[i] = l[i] * mul + 0.0f;
    }
    if (temperature < 0.01) { uint32_t r = argmax_first(l, vocab); free(l); return r; }
    if (top_k > 0) q3o_top_k_filter(l, vocab, top_k);
    if (top_p < 1.0 && top_p > 0.0) q3o_top_p_filter(l, vocab, top_p);
    float mx = -INFINITY, sum = 0.0f;
    for (int i = 0; i < vocab; ++i) if (l[i] > mx) mx = l[i];
    for (int i = 0; i < vocab; ++i) { l[i] = expf(l[i] - mx); sum += l[i]; }
    for (int i = 0; i < vocab; ++i) l[i] /= sum;
    float u = q3o_rng_next(rng_state);
    float cdf = 0.0f; uint32_t pick = 0; int found = 0;
    for (int i = 0; i < vocab; ++i) { cdf += l[i]; if (cdf >= u) { pick = (uint32_t)i; found = 1; break; } }
    if (!found) pick = 0;                                  /* argmin over all-(vocab+1) → 0 */
    free(l);
    return pick;
}

/* lib.rs:1417-1431 */
void q3o_codes_to_tensor(const uint32_t* frames, int n_frames, int64_t* out) {
    for (int f = 0; f < n_frames; ++f)
        for (int q = 0; q < 16; ++q) out[(size_t)q * n_frames + f] = (int64_t)frames[(size_t)f * 16 + q];
}

/* ------------------------------------------------------------------------------------------
 * talker pieces
 * ---------------------------------------------------------------------------------------- */
/* TextProjection::forward (talker.rs:316-320) over gathered text embeddings */
static void text_project(const q3o_model* m, const uint32_t* ids, int n, float* out /*[n][H]*/) {
    const q3o_config* c = &m->cfg;
    int TD = c->text_dim, H = c->hidden;
    if (n <= 0) return;
    float* e = fmalloc((size_t)n * TD);
    float* h = fmalloc((size_t)n * TD);
    for (int i = 0; i < n; ++i) memcpy(e + (size_t)i * TD, m->text_emb + (size_t)ids[i] * TD, (size_t)TD * sizeof(float));
    q3o_linear(e, m->fc1w, m->fc1b, h, n, TD, TD);
    for (size_t i = 0; i < (size_t)n * TD; ++i) h[i] = silu_f(h[i]);
    q3o_linear(h, m->fc2w, m->fc2b, out, n, H, TD);
    free(e); free(h);
}

struct q3o_session {
    q3o_model* m;
    q3o_request req;
    uint32_t* text_ids; uint32_t* instruct_ids; float* xvector; uint32_t* ref_codes; uint32_t* ref_text_ids;
    kvc_t* kv;           /* talker caches */
    kvc_t* cpkv;         /* code predictor caches */
    int prefill_len, offset;
    float* prefill_embeds;
    float* last_hidden;  /* [H] normed */
    float* logits;       /* [codec_vocab] */
    float* trailing; int trailing_len; float* pad_embed;
    uint64_t rng;
};

static void add_rows(float* dst, const float* a, const float* b, int n) { for (int i = 0; i < n; ++i) dst[i] = a[i] + b[i]; }

/* run_prefill_layers (talker.rs:823-841): S tokens at offset 0, norm all, head on last */
static void talker_prefill(q3o_session* s, float* hidden, int S) {
    const q3o_model* m = s->m; const q3o_config* c = &m->cfg;
    for (int i = 0; i < c->n_layers; ++i)
        layer_forward(&m->tl[i], hidden, S, c->hidden, c->inter, c->n_heads, c->n_kv_heads, c->head_dim,
                      c->rms_eps, c->rope_theta, &s->kv[i], 0);
    q3o_rms_norm(hidden, m->norm, hidden, S, c->hidden, c->rms_eps);
    memcpy(s->last_hidden, hidden + (size_t)(S - 1) * c->hidden, (size_t)c->hidden * sizeof(float));
    q3o_linear(s->last_hidden, m->codec_head, NULL, s->logits, 1, c->codec_vocab, c->hidden);
}

q3o_session* q3o_session_new(q3o_model* m, const q3o_request* req) {
    if (!m->have_lm) { fail("model not finalized for talker/code predictor"); return NULL; }
    const q3o_config* c = &m->cfg;
    int H = c->hidden;
    q3o_session* s = (q3o_session*)calloc(1, sizeof *s);
    s->m = m; s->req = *req;
    if (req->n_text > 0) { s->text_ids = (uint32_t*)malloc((size_t)req->n_text * 4); memcpy(s->text_ids, req->text_ids, (size_t)req->n_text * 4); }
    if (req->n_instruct > 0) { s->instruct_ids = (uint32_t*)malloc((size_t)req->n_instruct * 4); memcpy(s->instruct_ids, req->instruct_ids, (size_t)req->n_instruct * 4); }
    if (req->xvector) { s->xvector = fmalloc((size_t)H); memcpy(s->xvector, req->xvector, (size_t)H * sizeof(float)); }
    const int icl = req->mode == Q3O_MODE_VOICE_CLONE && req->n_ref > 0 && req->ref_codes && req->ref_text_ids;
    if (icl) {
        s->ref_codes = (uint32_t*)malloc((size_t)req->n_ref * 16 * 4); memcpy(s->ref_codes, req->ref_codes, (size_t)req->n_ref * 16 * 4);
        s->ref_text_ids = (uint32_t*)malloc((size_t)(req->n_ref_text > 0 ? req->n_ref_text : 1) * 4);
        if (req->n_ref_text > 0) memcpy(s->ref_text_ids, req->ref_text_ids, (size_t)req->n_ref_text * 4);
        /* lib.rs:913-929: ICL floors the repetition penalty at 1.5 and caps max_new_tokens */
        if (s->req.opts.repetition_penalty < 1.5) s->req.opts.repetition_penalty = 1.5;
        int cap = 6 * req->n_text; if (cap < 75) cap = 75;
        if (s->req.opts.max_length > cap) s->req.opts.max_length = cap;
    }
    s->req.text_ids = s->text_ids; s->req.instruct_ids = s->instruct_ids; s->req.xvector = s->xvector;
    s->req.ref_codes = s->ref_codes; s->req.ref_text_ids = s->ref_text_ids;
    s->kv = (kvc_t*)calloc((size_t)c->n_layers, sizeof(kvc_t));
    for (int i = 0; i < c->n_layers; ++i) kvc_init(&s->kv[i], c->n_kv_heads, c->head_dim, 64);
    s->cpkv = (kvc_t*)calloc((size_t)c->cp_layers, sizeof(kvc_t));
    for (int i = 0; i < c->cp_layers; ++i) kvc_init(&s->cpkv[i], c->cp_kv_heads, c->head_dim, 17);
    s->last_hidden = fmalloc((size_t)H);
    s->logits = fmalloc((size_t)c->codec_vocab);
    if (req->opts.has_seed) q3o_rng_seed(req->opts.seed, &s->rng);
    else q3o_rng_seed(0, &s->rng);

    /* build_trailing_text (lib.rs:508-519) */
    {
        int nt = req->n_text > 1 ? req->n_text - 1 : 0;
        s->trailing_len = nt + 1;
        s->trailing = fmalloc((size_t)s->trailing_len * H);
        if (nt > 0) text_project(m, s->text_ids + 1, nt, s->trailing);
        uint32_t eos = TTS_EOS; text_project(m, &eos, 1, s->trailing + (size_t)nt * H);
        s->pad_embed = fmalloc((size_t)H);
        uint32_t pad = TTS_PAD; text_project(m, &pad, 1, s->pad_embed);
    }

    /* prefill embeddings: talker.rs:451-491 (custom voice), 511-564 (voice clone, x-vector
     * only), 585-627 (voice design) */
    int mode = req->mode;
    int n_ins = mode == Q3O_MODE_VOICE_DESIGN ? req->n_instruct : 0;
    int n_codec_overlay = mode == Q3O_MODE_VOICE_DESIGN ? 5 : 6;
    int has_first = req->n_text > 0 && !icl;            /* icl_mode omits first_text + codec_bos (talker.rs:555-561) */
    int n_icl = icl ? req->n_ref + 1 : 0;                /* streaming overlay: icl_len = n_codec (talker.rs:690-709) */
    int S = n_ins + 3 + n_codec_overlay + (has_first ? 1 : 0) + n_icl;
    float* emb = fmalloc((size_t)S * H);
    float* row = emb;
    if (n_ins > 0) { text_project(m, s->instruct_ids, n_ins, row); row += (size_t)n_ins * H; }
    { uint32_t role[3] = { IM_START, ASSISTANT, NEWLINE }; text_project(m, role, 3, row); row += (size_t)3 * H; }
    float* pad_proj = fmalloc((size_t)H); float* bos_proj = fmalloc((size_t)H);
    { uint32_t t = TTS_PAD; text_project(m, &t, 1, pad_proj); t = TTS_BOS; text_project(m, &t, 1, bos_proj); }
    uint32_t codec_ids[7]; int n_codec;
    if (mode == Q3O_MODE_VOICE_DESIGN) {
        uint32_t ids[6] = { CODEC_THINK, CODEC_THINK_BOS, req->language_id, CODEC_THINK_EOS, CODEC_PAD, CODEC_BOS };
        memcpy(codec_ids, ids, sizeof ids); n_codec = 6;
    } else {
        uint32_t ids[7] = { CODEC_THINK, CODEC_THINK_BOS, req->language_id, CODEC_THINK_EOS, req->speaker_id, CODEC_PAD, CODEC_BOS };
        memcpy(codec_ids, ids, sizeof ids); n_codec = 7;
    }
    for (int i = 0; i < n_codec_overlay; ++i) {
        const float* txt = (i == n_codec_overlay - 1) ? bos_proj : pad_proj;
        const float* cod = m->codec_emb + (size_t)codec_ids[i] * H;
        if (mode == Q3O_MODE_VOICE_CLONE && i == 4) cod = s->xvector;
        add_rows(row, txt, cod, H); row += H;
    }
    if (has_first) {
        float* ft = fmalloc((size_t)H);
        text_project(m, s->text_ids, 1, ft);
        add_rows(row, ft, m->codec_emb + (size_t)codec_ids[n_codec - 1] * H, H);
        free(ft);
    }
    if (icl) {
        /* build_icl_prompt, streaming mode (talker.rs:646-709): text = proj([ref_text, target_text, tts_eos]);
         * codec = [codec_emb[BOS]; Σ16 ref embeddings per frame (lib.rs:1239-1257)]; overlay element-wise */
        if (has_first) row += H;
        else if (req->n_text > 0) { /* nothing: position 9 skipped */ }
        int n_text_all = req->n_ref_text + req->n_text + 1;
        uint32_t* ids = (uint32_t*)malloc((size_t)n_text_all * 4);
        memcpy(ids, s->ref_text_ids, (size_t)req->n_ref_text * 4);
        memcpy(ids + req->n_ref_text, s->text_ids, (size_t)req->n_text * 4);
        ids[n_text_all - 1] = TTS_EOS;
        float* tproj = fmalloc((size_t)n_text_all * H);
        text_project(m, ids, n_text_all, tproj);
        float* cod = fmalloc((size_t)H);
        float* icl_row = emb + (size_t)(S - n_icl) * H;
        for (int i = 0; i < n_icl; ++i) {
            if (i == 0) memcpy(cod, m->codec_emb + (size_t)CODEC_BOS * H, (size_t)H * sizeof(float));
            else {
                const uint32_t* fr = s->ref_codes + (size_t)(i - 1) * 16;
                memcpy(cod, m->codec_emb + (size_t)fr[0] * H, (size_t)H * sizeof(float));
                for (int g = 1; g < 16; ++g) {
                    const float* e = m->cp_emb[g - 1] + (size_t)fr[g] * H;
                    for (int k = 0; k < H; ++k) cod[k] = cod[k] + e[k];
                }
            }
            const float* txt = i < n_text_all ? tproj + (size_t)i * H : pad_proj;
            add_rows(icl_row + (size_t)i * H, txt, cod, H);
        }
        /* trailing text: remaining text rows, or tts_pad (talker.rs:692-708) */
        free(s->trailing);
        if (n_text_all > n_icl) {
            s->trailing_len = n_text_all - n_icl;
            s->trailing = fmalloc((size_t)s->trailing_len * H);
            memcpy(s->trailing, tproj + (size_t)n_icl * H, (size_t)s->trailing_len * H * sizeof(float));
        } else {
            s->trailing_len = 1;
            s->trailing = fmalloc((size_t)H);
            memcpy(s->trailing, pad_proj, (size_t)H * sizeof(float));
        }
        free(ids); free(tproj); free(cod);
    }
    free(pad_proj); free(bos_proj);
    s->prefill_len = S;
    s->prefill_embeds = fmalloc((size_t)S * H);
    memcpy(s->prefill_embeds, emb, (size_t)S * H * sizeof(float));
    talker_prefill(s, emb, S);
    s->offset = S;
    free(emb);
    return s;
}

void q3o_session_free(q3o_session* s) {
    if (!s) return;
    const q3o_config* c = &s->m->cfg;
    for (int i = 0; i < c->n_layers; ++i) kvc_free(&s->kv[i]);
    for (int i = 0; i < c->cp_layers; ++i) kvc_free(&s->cpkv[i]);
    free(s->kv); free(s->cpkv); free(s->text_ids); free(s->instruct_ids); free(s->xvector); free(s->ref_codes); free(s->ref_text_ids);
    free(s->prefill_embeds); free(s->last_hidden); free(s->logits); free(s->trailing); free(s->pad_embed);
    free(s);
}
int q3o_session_prefill_len(const q3o_session* s) { return s->prefill_len; }
void q3o_session_effective(const q3o_session* s, double* rp, int* ml) { if (rp) *rp = s->req.opts.repetition_penalty; if (ml) *ml = s->req.opts.max_length; }
void q3o_session_prefill_out(const q3o_session* s, float* last_hidden, float* logits) {
    memcpy(last_hidden, s->last_hidden, (size_t)s->m->cfg.hidden * sizeof(float));
    memcpy(logits, s->logits, (size_t)s->m->cfg.codec_vocab * sizeof(float));
}
void q3o_session_prefill_embeds(const q3o_session* s, float* out) {
    memcpy(out, s->prefill_embeds, (size_t)s->prefill_len * s->m->cfg.hidden * sizeof(float));
}
int q3o_session_trailing_len(const q3o_session* s) { return s->trailing_len; }
void q3o_session_trailing(const q3o_session* s, float* trailing, float* pad) {
    int H = s->m->cfg.hidden;
    if (trailing) memcpy(trailing, s->trailing, (size_t)s->trailing_len * H * sizeof(float));
    if (pad) memcpy(pad, s->pad_embed, (size_t)H * sizeof(float));
}

/* generate_step_with_embed (talker.rs:716-736) */
void q3o_session_talker_step(q3o_session* s, const float* input_embed, float* hidden_out, float* logits_out) {
    const q3o_model* m = s->m; const q3o_config* c = &m->cfg;
    int H = c->hidden;
    float* h = fmalloc((size_t)H);
    memcpy(h, input_embed, (size_t)H * sizeof(float));
    for (int i = 0; i < c->n_layers; ++i)
        layer_forward(&m->tl[i], h, 1, H, c->inter, c->n_heads, c->n_kv_heads, c->head_dim,
                      c->rms_eps, c->rope_theta, &s->kv[i], s->offset);
    s->offset += 1;
    q3o_rms_norm(h, m->norm, hidden_out, 1, H, c->rms_eps);
    q3o_linear(hidden_out, m->codec_head, NULL, logits_out, 1, c->codec_vocab, H);
    free(h);
}

/* CodePredictor::generate_acoustic_codes (code_predictor.rs:320-416) */
void q3o_session_cp_generate(q3o_session* s, const float* last_hidden, const float* sem_embed,
                             uint32_t* codes15, float* cp_logits) {
    const q3o_model* m = s->m; const q3o_config* c = &m->cfg;
    int H = c->hidden, CH = c->cp_hidden, V = c->cp_vocab, n_ac = c->n_groups - 1;
    for (int i = 0; i < c->cp_layers; ++i) s->cpkv[i].len = 0;           /* cache.reset() */
    float* in2 = fmalloc((size_t)2 * H);
    memcpy(in2, last_hidden, (size_t)H * sizeof(float));
    memcpy(in2 + H, sem_embed, (size_t)H * sizeof(float));
    float* hid = fmalloc((size_t)2 * CH);
    if (m->mtp_w) q3o_linear(in2, m->mtp_w, m->mtp_b, hid, 2, CH, H);
    else memcpy(hid, in2, (size_t)2 * CH * sizeof(float));
    for (int i = 0; i < c->cp_layers; ++i)
        layer_forward(&m->cl[i], hid, 2, CH, c->cp_inter, c->cp_heads, c->cp_kv_heads, c->head_dim,
                      c->rms_eps, c->rope_theta, &s->cpkv[i], 0);
    q3o_rms_norm(hid, m->cp_norm, hid, 2, CH, c->rms_eps);
    float* lg = fmalloc((size_t)V);
    q3o_linear(hid + CH, m->cp_head[0], NULL, lg, 1, V, CH);
    if (cp_logits) memcpy(cp_logits, lg, (size_t)V * sizeof(float));
    uint32_t prev = argmax_first(lg, V);
    codes15[0] = prev;
    int offset = 2;
    float* h = fmalloc((size_t)CH);
    for (int g = 1; g < n_ac; ++g) {
        const float* e = m->cp_emb[g - 1] + (size_t)prev * H;
        if (m->mtp_w) q3o_linear(e, m->mtp_w, m->mtp_b, h, 1, CH, H);
        else memcpy(h, e, (size_t)CH * sizeof(float));
        for (int i = 0; i < c->cp_layers; ++i)
            layer_forward(&m->cl[i], h, 1, CH, c->cp_inter, c->cp_heads, c->cp_kv_heads, c->head_dim,
                          c->rms_eps, c->rope_theta, &s->cpkv[i], offset);
        q3o_rms_norm(h, m->cp_norm, h, 1, CH, c->rms_eps);
        q3o_linear(h, m->cp_head[g], NULL, lg, 1, V, CH);
        if (cp_logits) memcpy(cp_logits + (size_t)g * V, lg, (size_t)V * sizeof(float));
        prev = argmax_first(lg, V);
        codes15[g] = prev;
        offset += 1;
    }
    free(in2); free(hid); free(lg); free(h);
}

/* lib.rs:588-590, 612-622 + code_predictor.rs:497-519: ((e0+e1)+...+e14); sem + that; + text */
void q3o_frame_embed(q3o_model* m, uint32_t sem_token, const uint32_t* codes15, const float* text_add, float* out) {
    int H = m->cfg.hidden, n_ac = m->cfg.n_groups - 1;
    const float* sem = m->codec_emb + (size_t)sem_token * H;
    float* acc = fmalloc((size_t)H);
    memcpy(acc, m->cp_emb[0] + (size_t)codes15[0] * H, (size_t)H * sizeof(float));
    for (int g = 1; g < n_ac; ++g) {
        const float* e = m->cp_emb[g] + (size_t)codes15[g] * H;
        for (int i = 0; i < H; ++i) acc[i] = acc[i] + e[i];
    }
    for (int i = 0; i < H; ++i) { float summed = sem[i] + acc[i]; out[i] = summed + text_add[i]; }
    free(acc);
}

/* generate_codes (lib.rs:530-656) */
int q3o_session_generate(q3o_session* s, uint32_t* codes_out, float* talker_logits, float* cp_logits) {
    q3o_model* m = s->m; const q3o_config* c = &m->cfg; const q3o_options* o = &s->req.opts;
    int H = c->hidden, V = c->codec_vocab, CV = c->cp_vocab, n_ac = c->n_groups - 1;
    uint8_t* seen = (uint8_t*)calloc((size_t)V, 1);
    float* lg = fmalloc((size_t)V);
    float* step_in = fmalloc((size_t)H);
    float* new_logits = fmalloc((size_t)V);
    float* hid = fmalloc((size_t)H);
    memcpy(lg, s->logits, (size_t)V * sizeof(float));
    if (talker_logits) memcpy(talker_logits, lg, (size_t)V * sizeof(float));
    q3o_apply_penalties(lg, V, seen, o->repetition_penalty, 0, o->min_new_tokens, o->eos_token_id);
    uint32_t tok = q3o_sample(lg, V, o->temperature, o->top_k, o->top_p, &s->rng);
    if ((int)tok < V) seen[tok] = 1;
    int token_count = 1, n_frames = 0;
    for (int f = 0; f < o->max_length; ++f) {
        if (o->eos_token_id >= 0 && (int)tok == o->eos_token_id) break;
        const float* sem = m->codec_emb + (size_t)tok * H;
        uint32_t* frame = codes_out + (size_t)f * 16;
        frame[0] = tok;
        q3o_session_cp_generate(s, s->last_hidden, sem, frame + 1, cp_logits ? cp_logits + (size_t)f * n_ac * CV : NULL);
        n_frames = f + 1;
        const float* text_add = f < s->trailing_len ? s->trailing + (size_t)f * H : s->pad_embed;
        q3o_frame_embed(m, tok, frame + 1, text_add, step_in);
        q3o_session_talker_step(s, step_in, hid, new_logits);
        memcpy(s->last_hidden, hid, (size_t)H * sizeof(float));
        if (talker_logits) memcpy(talker_logits + (size_t)(f + 1) * V, new_logits, (size_t)V * sizeof(float));
        memcpy(lg, new_logits, (size_t)V * sizeof(float));
        q3o_apply_penalties(lg, V, seen, o->repetition_penalty, token_count, o->min_new_tokens, o->eos_token_id);
        tok = q3o_sample(lg, V, o->temperature, o->top_k, o->top_p, &s->rng);
        if ((int)tok < V) seen[tok] = 1;
        token_count += 1;
    }
    free(seen); free(lg); free(step_in); free(new_logits); free(hid);
    return n_frames;
}

/* ------------------------------------------------------------------------------------------
 * codec decoder (src/models/codec). Conv tensors are [C][L] row-major (time contiguous).
 * ---------------------------------------------------------------------------------------- */
/* CausalConv1d::forward (causal_conv.rs:94-103): left zero-pad dil*(k-1), stride 1.
 * weight [cout][cin/groups][k]. */
void q3o_causal_conv1d(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L,
                       int k, int dil, int groups) {
    int cin_g = cin / groups, cout_g = cout / groups;
#pragma omp parallel for schedule(static) PAR_IF((double)cout * cin_g * k * L)
    for (int co = 0; co < cout; ++co) {
        float* yr = y + (size_t)co * L;
        for (int t = 0; t < L; ++t) yr[t] = 0.0f;
        int g = co / cout_g;
        for (int ci = 0; ci < cin_g; ++ci) {
            const float* xr = x + (size_t)(g * cin_g + ci) * L;
            const float* wr = w + ((size_t)co * cin_g + ci) * k;
            for (int kk = 0; kk < k; ++kk) {
                int sh = (k - 1 - kk) * dil;          /* y[t] += w[kk] * x[t - sh] */
                float wv = wr[kk];
                for (int t = sh; t < L; ++t) yr[t] += wv * xr[t - sh];
            }
        }
        if (b) { float bv = b[co]; for (int t = 0; t < L; ++t) yr[t] += bv; }
    }
}

/* CausalTransConv1d::forward (causal_trans_conv.rs:88-100): ConvTranspose1d(stride, pad 0),
 * weight [cin][cout][k], full length (L-1)*s + k, then drop the last k - s samples. */
void q3o_causal_trans_conv1d(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L,
                             int k, int stride) {
    int trim = k > stride ? k - stride : 0;
    int Lout = (L - 1) * stride + k - trim;
#pragma omp parallel for schedule(static) PAR_IF((double)cout * cin * k * L)
    for (int co = 0; co < cout; ++co) {
        float* yr = y + (size_t)co * Lout;
        for (int t = 0; t < Lout; ++t) yr[t] = 0.0f;
        for (int ci = 0; ci < cin; ++ci) {
            const float* xr = x + (size_t)ci * L;
            const float* wr = w + ((size_t)ci * cout + co) * k;
            for (int j = 0; j < L; ++j) {
                float xv = xr[j];
                for (int kk = 0; kk < k; ++kk) {
                    int t = j * stride + kk;
                    if (t < Lout) yr[t] += xv * wr[kk];
                }
            }
        }
        if (b) { float bv = b[co]; for (int t = 0; t < Lout; ++t) yr[t] += bv; }
    }
}

/* SnakeBeta::forward (snake_beta.rs:58-77): x + sin²(x·exp(α)) · 1/(exp(β)+1e-9) */
void q3o_snake_beta(const float* x, const float* alpha, const float* beta, float* y, int C, int L) {
#pragma omp parallel for schedule(static) PAR_IF((double)C * L * 20)
    for (int c = 0; c < C; ++c) {
        float a = expf(alpha[c]);
        float ib = 1.0f / (expf(beta[c]) + (float)1e-9);
        for (int t = 0; t < L; ++t) {
            float xv = x[(size_t)c * L + t];
            float sv = sinf(xv * a);
            y[(size_t)c * L + t] = xv + (sv * sv) * ib;
        }
    }
}

static void transpose2d(const float* x, float* y, int R, int C) {   /* [R][C] -> [C][R] */
    for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) y[(size_t)c * R + r] = x[(size_t)r * C + c];
}

/* Decoder12Hz::run_layer (decoder_12hz.rs:586-672). hidden [T][DH] in place. */
static void dec_layer_forward(const q3o_config* c, const dec_layer_w* L, float* hidden, int T,
                              const float* cs, const float* sn) {
    int DH = c->dec_hidden, nh = c->dec_heads, hd = c->dec_head_dim, QD = nh * hd, DI = c->dec_inter, half = hd / 2;
    float* nrm = fmalloc((size_t)T * DH);
    float* q = fmalloc((size_t)T * QD); float* k = fmalloc((size_t)T * QD); float* v = fmalloc((size_t)T * QD);
    float* att = fmalloc((size_t)T * QD);
    /* rms_norm (decoder_12hz.rs:675-679): x / sqrt(mean(x²)+eps) * w */
    for (int t = 0; t < T; ++t) {
        const float* xr = hidden + (size_t)t * DH; float ss = 0.0f;
        for (int i = 0; i < DH; ++i) ss += xr[i] * xr[i];
        float d = sqrtf(ss / (float)DH + c->dec_eps);
        for (int i = 0; i < DH; ++i) nrm[(size_t)t * DH + i] = xr[i] / d * L->in_ln[i];
    }
    q3o_linear(nrm, L->q, NULL, q, T, QD, DH);
    q3o_linear(nrm, L->k, NULL, k, T, QD, DH);
    q3o_linear(nrm, L->v, NULL, v, T, QD, DH);
    /* apply_rope (decoder_12hz.rs:682-691): x*cos + rotate_half(x)*sin, cos/sin = repeat(1,2) */
    for (int t = 0; t < T; ++t)
        for (int hh = 0; hh < 2 * nh; ++hh) {
            float* p = (hh < nh ? q : k) + (size_t)t * QD + (size_t)(hh % nh) * hd;
            float tmp[256];
            for (int i = 0; i < half; ++i) {
                float x1 = p[i], x2 = p[i + half], cv = cs[(size_t)t * half + i], sv = sn[(size_t)t * half + i];
                float a = x1 * cv, b = (-x2) * sv, d = x2 * cv, e = x1 * sv;
                tmp[i] = a + b; tmp[i + half] = d + e;
            }
            memcpy(p, tmp, (size_t)hd * sizeof(float));
        }
    float scale = (float)pow((double)hd, -0.5);
#pragma omp parallel for collapse(2) schedule(static) PAR_IF((double)T * T * nh * hd)
    for (int t = 0; t < T; ++t)
        for (int hh = 0; hh < nh; ++hh) {
            float* sc = fmalloc((size_t)t + 1);
            const float* qh = q + (size_t)t * QD + (size_t)hh * hd;
            float mx = -INFINITY;
            for (int j = 0; j <= t; ++j) {
                sc[j] = dot_f32(qh, k + (size_t)j * QD + (size_t)hh * hd, hd) * scale;
                if (sc[j] > mx) mx = sc[j];
            }
            float sum = 0.0f;
            for (int j = 0; j <= t; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
            for (int j = 0; j <= t; ++j) sc[j] /= sum;
            float* o = att + (size_t)t * QD + (size_t)hh * hd;
            for (int d = 0; d < hd; ++d) o[d] = 0.0f;
            for (int j = 0; j <= t; ++j) {
                float pj = sc[j]; const float* vj = v + (size_t)j * QD + (size_t)hh * hd;
                for (int d = 0; d < hd; ++d) o[d] += pj * vj[d];
            }
            free(sc);
        }
    float* ao = fmalloc((size_t)T * DH);
    q3o_linear(att, L->o, NULL, ao, T, DH, QD);
    for (int t = 0; t < T; ++t) for (int i = 0; i < DH; ++i) {
        size_t ix = (size_t)t * DH + i;
        hidden[ix] = hidden[ix] + ao[ix] * L->attn_scale[i];
    }
    for (int t = 0; t < T; ++t) {
        const float* xr = hidden + (size_t)t * DH; float ss = 0.0f;
        for (int i = 0; i < DH; ++i) ss += xr[i] * xr[i];
        float d = sqrtf(ss / (float)DH + c->dec_eps);
        for (int i = 0; i < DH; ++i) nrm[(size_t)t * DH + i] = xr[i] / d * L->post_ln[i];
    }
    float* g = fmalloc((size_t)T * DI); float* u = fmalloc((size_t)T * DI);
    q3o_linear(nrm, L->gate, NULL, g, T, DI, DH);
    q3o_linear(nrm, L->up, NULL, u, T, DI, DH);
    for (size_t i = 0; i < (size_t)T * DI; ++i) g[i] = silu_f(g[i]) * u[i];
    q3o_linear(g, L->down, NULL, ao, T, DH, DI);
    for (int t = 0; t < T; ++t) for (int i = 0; i < DH; ++i) {
        size_t ix = (size_t)t * DH + i;
        hidden[ix] = hidden[ix] + ao[ix] * L->mlp_scale[i];
    }
    free(nrm); free(q); free(k); free(v); free(att); free(ao); free(g); free(u);
}

/* ConvNeXtBlock::forward (convnext_block.rs:110-141). x [C][L] → y [C][L] */
static void convnext_forward(const upsample_w* U, const float* x, float* y, int C, int L) {
    float* dw = fmalloc((size_t)C * L);
    q3o_causal_conv1d(x, U->dww, U->dwb, dw, C, C, L, 7, 1, C);
    float* t1 = fmalloc((size_t)L * C);
    transpose2d(dw, t1, C, L);                      /* [L][C] */
    /* LayerNorm eps 1e-6 (candle layer_norm op: mean, var = E[x²]-mean²) */
    for (int t = 0; t < L; ++t) {
        float* r = t1 + (size_t)t * C; float s = 0.0f, s2 = 0.0f;
        for (int i = 0; i < C; ++i) { s += r[i]; s2 += r[i] * r[i]; }
        float mean = s / (float)C, var = s2 / (float)C - mean * mean;
        float inv = 1.0f / sqrtf(var + 1e-6f);
        for (int i = 0; i < C; ++i) r[i] = (r[i] - mean) * inv * U->nw[i] + U->nb[i];
    }
    float* h = fmalloc((size_t)L * 4 * C);
    q3o_linear(t1, U->p1w, U->p1b, h, L, 4 * C, C);
    for (size_t i = 0; i < (size_t)L * 4 * C; ++i) { float v = h[i]; h[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
    q3o_linear(h, U->p2w, U->p2b, t1, L, C, 4 * C);
    for (int t = 0; t < L; ++t) for (int i = 0; i < C; ++i) t1[(size_t)t * C + i] *= U->gamma[i];
    transpose2d(t1, dw, L, C);                      /* back to [C][L] */
    for (size_t i = 0; i < (size_t)C * L; ++i) y[i] = x[i] + dw[i];
    free(dw); free(t1); free(h);
}

static void tap(float** taps, int id, const float* data, size_t n) {
    if (taps && taps[id]) memcpy(taps[id], data, n * sizeof(float));
}

int q3o_decode_taps(q3o_model* m, const int64_t* codes, int T, float* pcm, float** taps) {
    if (!m->have_dec) return fail("model not finalized for decoder");
    const q3o_config* c = &m->cfg;
    int CD = c->dec_cb_dim, Q = c->dec_q_dim, LAT = c->dec_latent, DH = c->dec_hidden, CB = c->dec_cb_size;
    if (T <= 0) return 0;
    /* 1. quantizer decode (decoder_12hz.rs:420-452) */
    float* fe = fmalloc((size_t)T * CD); float* re = fcalloc((size_t)T * CD);
    for (int t = 0; t < T; ++t) {
        int64_t c0 = codes[t] % CB;
        memcpy(fe + (size_t)t * CD, m->first_cb + (size_t)c0 * CD, (size_t)CD * sizeof(float));
    }
    for (int i = 0; i < 15; ++i)
        for (int t = 0; t < T; ++t) {
            int64_t ci = codes[(size_t)(i + 1) * T + t];
            if (ci < 0 || ci >= CB) { free(fe); free(re); return fail("code %lld out of range for codebook %d", (long long)ci, i + 1); }
            const float* e = m->rest_cb[i] + (size_t)ci * CD;
            for (int d = 0; d < CD; ++d) re[(size_t)t * CD + d] = re[(size_t)t * CD + d] + e[d];
        }
    float* fp = fmalloc((size_t)T * Q); float* rp = fmalloc((size_t)T * Q);
    q3o_linear(fe, m->first_proj, NULL, fp, T, Q, CD);
    q3o_linear(re, m->rest_proj, NULL, rp, T, Q, CD);
    for (size_t i = 0; i < (size_t)T * Q; ++i) fp[i] = fp[i] + rp[i];
    float* quant = fmalloc((size_t)Q * T);
    transpose2d(fp, quant, T, Q);                    /* [Q][T] */
    tap(taps, Q3O_DEC_QUANT, quant, (size_t)Q * T);
    free(fe); free(re); free(fp); free(rp);
    /* 2. pre_conv k=3 */
    float* pc = fmalloc((size_t)LAT * T);
    q3o_causal_conv1d(quant, m->pre_w, m->pre_b, pc, Q, LAT, T, 3, 1, 1);
    tap(taps, Q3O_DEC_PRECONV, pc, (size_t)LAT * T);
    free(quant);
    /* 3. pre-transformer */
    float* ht = fmalloc((size_t)T * LAT);
    transpose2d(pc, ht, LAT, T);                     /* [T][LAT] */
    float* hid = fmalloc((size_t)T * DH);
    q3o_linear(ht, m->inp_w, m->inp_b, hid, T, DH, LAT);
    int half = c->dec_head_dim / 2;
    float* cs = fmalloc((size_t)T * half); float* sn = fmalloc((size_t)T * half);
    q3o_rope_table(c->dec_theta, c->dec_head_dim, 0, T, cs, sn);
    for (int l = 0; l < c->dec_layers; ++l) dec_layer_forward(c, &m->dl[l], hid, T, cs, sn);
    for (int t = 0; t < T; ++t) {
        float* xr = hid + (size_t)t * DH; float ss = 0.0f;
        for (int i = 0; i < DH; ++i) ss += xr[i] * xr[i];
        float d = sqrtf(ss / (float)DH + c->dec_eps);
        for (int i = 0; i < DH; ++i) xr[i] = xr[i] / d * m->dec_norm[i];
    }
    q3o_linear(hid, m->outp_w, m->outp_b, ht, T, LAT, DH);
    transpose2d(ht, pc, T, LAT);                     /* [LAT][T] */
    tap(taps, Q3O_DEC_PRETRANS, pc, (size_t)LAT * T);
    free(ht); free(hid); free(cs); free(sn);
    /* 5. upsample stages */
    float* cur = pc; int L = T;
    for (int i = 0; i < 2; ++i) {
        int r = c->dec_up_ratios[i];
        float* up = fmalloc((size_t)LAT * L * r);
        q3o_causal_trans_conv1d(cur, m->up[i].tw, m->up[i].tb, up, LAT, LAT, L, r, r);
        L *= r;
        float* cn = fmalloc((size_t)LAT * L);
        convnext_forward(&m->up[i], up, cn, LAT, L);
        free(up); free(cur); cur = cn;
        tap(taps, Q3O_DEC_UP0 + i, cur, (size_t)LAT * L);
    }
    /* 6. decoder.0 */
    int C = c->dec_dim;
    float* x = fmalloc((size_t)C * L);
    q3o_causal_conv1d(cur, m->init_w, m->init_b, x, LAT, C, L, 7, 1, 1);
    tap(taps, Q3O_DEC_INIT, x, (size_t)C * L);
    free(cur);
    /* 7. decoder blocks (decoder_block.rs:240-247, 81-92) */
    static const int dils[3] = { 1, 3, 9 };
    for (int b = 0; b < 4; ++b) {
        const dec_block_w* B = &m->blk[b];
        int r = c->dec_up_rates[b], Co = C / 2;
        float* sx = fmalloc((size_t)C * L);
        q3o_snake_beta(x, B->alpha, B->beta, sx, C, L);
        float* y = fmalloc((size_t)Co * L * r);
        q3o_causal_trans_conv1d(sx, B->tw, B->tb, y, C, Co, L, 2 * r, r);
        free(sx); free(x);
        L *= r; C = Co;
        float* t1 = fmalloc((size_t)C * L); float* t2 = fmalloc((size_t)C * L);
        for (int u = 0; u < 3; ++u) {
            const res_unit_w* R = &B->res[u];
            q3o_snake_beta(y, R->a1, R->b1, t1, C, L);
            q3o_causal_conv1d(t1, R->c1w, R->c1b, t2, C, C, L, 7, dils[u], 1);
            q3o_snake_beta(t2, R->a2, R->b2, t1, C, L);
            q3o_causal_conv1d(t1, R->c2w, R->c2b, t2, C, C, L, 1, 1, 1);
            for (size_t i = 0; i < (size_t)C * L; ++i) y[i] = t2[i] + y[i];
        }
        free(t1); free(t2);
        x = y;
        tap(taps, Q3O_DEC_BLK0 + b, x, (size_t)C * L);
    }
    /* 8-10. final snake, conv k7 → 1 channel, clamp */
    float* sx = fmalloc((size_t)C * L);
    q3o_snake_beta(x, m->fin_a, m->fin_b, sx, C, L);
    q3o_causal_conv1d(sx, m->fin_w, m->fin_bias, pcm, C, 1, L, 7, 1, 1);
    for (int t = 0; t < L; ++t) { float v = pcm[t]; pcm[t] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v); }
    free(sx); free(x);
    return L;
}

int q3o_decode(q3o_model* m, const int64_t* codes, int T, float* pcm) {
    return q3o_decode_taps(m, codes, T, pcm, NULL);
}
