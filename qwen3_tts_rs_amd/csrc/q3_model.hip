// q3_model.hip — errors, synthetic tensors, the weight manifest and arena, q3_model_* (create / set_tensor / finalize / KV pool API)
// (one of the five units of the engine: q3_engine.h says which holds what)
#include "q3_engine.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
q3_status set_err(q3_status st, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return st;
}
extern "C" q3_status q3i_set_err(q3_status st, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return st;
}

extern "C" int q3_abi_version(void) { return Q3_ABI_VERSION; }
extern "C" const char* q3_last_error(void) { return g_err; }
extern "C" int q3_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// ------------------------------------------------------------------------------------------------
// synthetic tensor generator (host)
// ------------------------------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static inline uint64_t fnv1a64(const char* s) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 0x100000001b3ULL; }
    return h;
}
extern "C" q3_status q3_synth_fill(uint64_t seed, const char* name, int dtype, float scale, float offset, int64_t n,
                                   void* out_host) {
    if (!name || !out_host || n < 0) return set_err(Q3_INVALID_ARG, "q3_synth_fill: bad argument");
    const uint64_t key = splitmix64(seed ^ fnv1a64(name));
    // Irwin-Hall(4) of 16-bit uniforms: exact integer sum, std = 65536/sqrt(3)
    const float c = (float)((double)scale / (65536.0 / 1.7320508075688772));
    float* of = (float*)out_host; uint16_t* ob = (uint16_t*)out_host;
    auto body = [=](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const uint64_t z = splitmix64(key + (uint64_t)i * 0x9E3779B97F4A7C15ULL);
            const int s = (int)(z & 0xffff) + (int)((z >> 16) & 0xffff) + (int)((z >> 32) & 0xffff) + (int)((z >> 48) & 0xffff) - 131070;
            const float v = offset + (float)s * c;
            if (dtype == Q3_DTYPE_BF16) ob[i] = f32_to_bf16_host(v);
            else of[i] = v;
        }
    };
    // plain std::thread fan-out (no OpenMP runtime inside the product library)
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 2 || n < (1 << 20)) { body(0, n); return Q3_OK; }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t lo = (int64_t)t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) th.emplace_back(body, lo, hi);
    }
    for (auto& t : th) t.join();
    return Q3_OK;
}

// PCG stream (sampling.rs:32-51, 84-94)
extern "C" void q3_rng_seed(uint64_t seed, uint64_t* state) { *state = seed * 2685821657736338717ULL + 1442695040888963407ULL; }
extern "C" float q3_rng_next(uint64_t* state) {
    const uint64_t old = *state;
    *state = old * 6364136223846793005ULL + 1442695040888963407ULL;
    const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
    const uint32_t out = (xs >> rot) | (xs << ((32 - rot) & 31));
    return (float)out / (float)UINT32_MAX;
}
extern "C" void q3_codes_to_tensor(const uint32_t* frames, int n_frames, int64_t* out) {
    for (int f = 0; f < n_frames; ++f)
        for (int q = 0; q < 16; ++q) out[(size_t)q * n_frames + f] = (int64_t)frames[(size_t)f * 16 + q];
}

static void add_slot(q3_model* m, const std::string& name, int64_t n, int stored, bool align = true) {
    Slot s; s.name = name; s.n = n; s.stored = stored;
    size_t off = m->arena_bytes;
    if (align) off = (off + 255) & ~(size_t)255;
    s.offset = off;
    m->arena_bytes = off + (size_t)n * (stored == Q3_DTYPE_BF16 ? 2 : 4);
    m->index[name] = (int)m->slots.size();
    m->slots.push_back(s);
}
// GEMV weight [rows][cols] bf16, stored MFMA-tiled (q3_kernels_gemv.hip); element count reported to the
// caller stays rows*cols (the checkpoint's), the arena holds the padded tiled image.
static void add_tiled(q3_model* m, const std::string& name, int rows, int cols, bool dual, bool align = true) {
    Slot s; s.name = name; s.n = (int64_t)rows * cols; s.stored = Q3_DTYPE_BF16; s.kind = SK_TILED; s.rows = rows; s.cols = cols; s.dual = dual;
    size_t off = m->arena_bytes;
    if (align) off = (off + 255) & ~(size_t)255;
    s.offset = off;
    m->arena_bytes = off + tiled_elems(1, rows, cols) * 2;
    m->index[name] = (int)m->slots.size();
    m->slots.push_back(s);
}
// second (4-row-tile) images of a group of tensors, laid out back to back (fused QKV needs them contiguous)
static void add_alt_images(q3_model* m, std::initializer_list<std::string> names) {
    bool first = true;
    for (const auto& nme : names) {
        Slot& s = m->slots[m->index[nme]];
        if (!s.dual) continue;
        size_t off = m->arena_bytes;
        if (first) off = (off + 255) & ~(size_t)255;
        first = false;
        s.offset2 = off;
        m->arena_bytes = off + tiled_elems(2, s.rows, s.cols) * 2;
    }
}
static void add_layer_slots(q3_model* m, const std::string& p, int H, int I, int nh, int nkv, int hd) {
    add_slot(m, p + ".input_layernorm.weight", H, Q3_DTYPE_F32);
    // q,k,v rows are stored back to back so the fused QKV GEMV sees one [QD+2KD][H] matrix
    const bool dq = dual_tiled((nh + 2 * nkv) * hd);
    add_tiled(m, p + ".self_attn.q_proj.weight", nh * hd, H, dq);
    add_tiled(m, p + ".self_attn.k_proj.weight", nkv * hd, H, dq, false);
    add_tiled(m, p + ".self_attn.v_proj.weight", nkv * hd, H, dq, false);
    add_alt_images(m, {p + ".self_attn.q_proj.weight", p + ".self_attn.k_proj.weight", p + ".self_attn.v_proj.weight"});
    add_tiled(m, p + ".self_attn.o_proj.weight", H, nh * hd, dual_tiled(H));
    add_alt_images(m, {p + ".self_attn.o_proj.weight"});
    add_slot(m, p + ".self_attn.q_norm.weight", hd, Q3_DTYPE_F32);
    add_slot(m, p + ".self_attn.k_norm.weight", hd, Q3_DTYPE_F32);
    add_slot(m, p + ".post_attention_layernorm.weight", H, Q3_DTYPE_F32);
    add_tiled(m, p + ".mlp.gate_proj.weight", I, H, dual_tiled(I));
    add_tiled(m, p + ".mlp.up_proj.weight", I, H, dual_tiled(I));
    add_tiled(m, p + ".mlp.down_proj.weight", H, I, dual_tiled(H));
    add_alt_images(m, {p + ".mlp.gate_proj.weight"}); add_alt_images(m, {p + ".mlp.up_proj.weight"}); add_alt_images(m, {p + ".mlp.down_proj.weight"});
}
static std::string fmt(const char* f, ...) {
    char b[256]; va_list ap; va_start(ap, f); vsnprintf(b, sizeof b, f, ap); va_end(ap); return b;
}

// tensor manifest: names/shapes of SURVEY.md Appendix B (talker.rs:380-405, code_predictor.rs:163-205,
// decoder_12hz.rs:191-381)
static void build_manifest(q3_model* m) {
    const q3_config& c = m->cfg;
    const int H = c.hidden, TD = c.text_dim, CH = c.cp_hidden;
    add_slot(m, "talker.model.text_embedding.weight", (int64_t)c.text_vocab * TD, Q3_DTYPE_BF16);
    add_tiled(m, "talker.text_projection.linear_fc1.weight", TD, TD, dual_tiled(TD)); add_alt_images(m, {"talker.text_projection.linear_fc1.weight"});
    add_slot(m, "talker.text_projection.linear_fc1.bias", TD, Q3_DTYPE_F32);
    add_tiled(m, "talker.text_projection.linear_fc2.weight", H, TD, dual_tiled(H)); add_alt_images(m, {"talker.text_projection.linear_fc2.weight"});
    add_slot(m, "talker.text_projection.linear_fc2.bias", H, Q3_DTYPE_F32);
    add_slot(m, "talker.model.codec_embedding.weight", (int64_t)c.codec_vocab * H, Q3_DTYPE_BF16);
    for (int i = 0; i < c.n_layers; ++i)
        add_layer_slots(m, fmt("talker.model.layers.%d", i), H, c.inter, c.n_heads, c.n_kv_heads, c.head_dim);
    add_slot(m, "talker.model.norm.weight", H, Q3_DTYPE_F32);
    add_tiled(m, "talker.codec_head.weight", c.codec_vocab, H, dual_tiled(c.codec_vocab)); add_alt_images(m, {"talker.codec_head.weight"});
    if (H != CH) {
        add_tiled(m, "talker.code_predictor.small_to_mtp_projection.weight", CH, H, dual_tiled(CH)); add_alt_images(m, {"talker.code_predictor.small_to_mtp_projection.weight"});
        add_slot(m, "talker.code_predictor.small_to_mtp_projection.bias", CH, Q3_DTYPE_F32);
    }
    for (int g = 0; g < c.n_groups - 1; ++g)
        add_slot(m, fmt("talker.code_predictor.model.codec_embedding.%d.weight", g), (int64_t)c.cp_vocab * H, Q3_DTYPE_BF16);
    for (int i = 0; i < c.cp_layers; ++i)
        add_layer_slots(m, fmt("talker.code_predictor.model.layers.%d", i), CH, c.cp_inter, c.cp_heads, c.cp_kv_heads, c.head_dim);
    add_slot(m, "talker.code_predictor.model.norm.weight", CH, Q3_DTYPE_F32);
    for (int g = 0; g < c.n_groups - 1; ++g)
        { const std::string nm = fmt("talker.code_predictor.lm_head.%d.weight", g); add_tiled(m, nm, c.cp_vocab, CH, dual_tiled(c.cp_vocab)); add_alt_images(m, {nm}); }
    // decoder (all f32)
    const int CB = c.dec_cb_size, CD = c.dec_cb_dim, Q = c.dec_q_dim, LAT = c.dec_latent, DH = c.dec_hidden;
    const int QD = c.dec_heads * c.dec_head_dim, DI = c.dec_inter;
    auto F = [&](const std::string& n, int64_t cnt) { add_slot(m, n, cnt, Q3_DTYPE_F32); };
    F("decoder.quantizer.rvq_first.vq.layers.0._codebook.embedding_sum", (int64_t)CB * CD);
    F("decoder.quantizer.rvq_first.vq.layers.0._codebook.cluster_usage", CB);
    for (int i = 0; i < 15; ++i) {
        F(fmt("decoder.quantizer.rvq_rest.vq.layers.%d._codebook.embedding_sum", i), (int64_t)CB * CD);
        F(fmt("decoder.quantizer.rvq_rest.vq.layers.%d._codebook.cluster_usage", i), CB);
    }
    F("decoder.quantizer.rvq_first.output_proj.weight", (int64_t)Q * CD);
    F("decoder.quantizer.rvq_rest.output_proj.weight", (int64_t)Q * CD);
    F("decoder.pre_conv.conv.weight", (int64_t)LAT * Q * 3);
    F("decoder.pre_conv.conv.bias", LAT);
    F("decoder.pre_transformer.input_proj.weight", (int64_t)DH * LAT);
    F("decoder.pre_transformer.input_proj.bias", DH);
    F("decoder.pre_transformer.output_proj.weight", (int64_t)LAT * DH);
    F("decoder.pre_transformer.output_proj.bias", LAT);
    F("decoder.pre_transformer.norm.weight", DH);
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = fmt("decoder.pre_transformer.layers.%d", i);
        F(p + ".input_layernorm.weight", DH);
        F(p + ".self_attn.q_proj.weight", (int64_t)QD * DH);
        F(p + ".self_attn.k_proj.weight", (int64_t)QD * DH);
        F(p + ".self_attn.v_proj.weight", (int64_t)QD * DH);
        F(p + ".self_attn.o_proj.weight", (int64_t)DH * QD);
        F(p + ".self_attn_layer_scale.scale", DH);
        F(p + ".post_attention_layernorm.weight", DH);
        F(p + ".mlp.gate_proj.weight", (int64_t)DI * DH);
        F(p + ".mlp.up_proj.weight", (int64_t)DI * DH);
        F(p + ".mlp.down_proj.weight", (int64_t)DH * DI);
        F(p + ".mlp_layer_scale.scale", DH);
    }
    auto TC = [&](const std::string& n, int cin, int cout, int k, int stride) {
        add_slot(m, n, (int64_t)cin * cout * k, Q3_DTYPE_F32);
        Slot& s = m->slots.back(); s.kind = SK_TRANSCONV; s.tc_cin = cin; s.tc_cout = cout; s.tc_k = k; s.tc_stride = stride;
    };
    for (int i = 0; i < 2; ++i) {
        const std::string p = fmt("decoder.upsample.%d", i);
        const int r = c.dec_up_ratios[i];
        TC(p + ".0.conv.weight", LAT, LAT, r, r);
        F(p + ".0.conv.bias", LAT);
        F(p + ".1.dwconv.conv.weight", (int64_t)LAT * 7);
        F(p + ".1.dwconv.conv.bias", LAT);
        F(p + ".1.norm.weight", LAT);
        F(p + ".1.norm.bias", LAT);
        F(p + ".1.pwconv1.weight", (int64_t)4 * LAT * LAT);
        F(p + ".1.pwconv1.bias", 4 * LAT);
        F(p + ".1.pwconv2.weight", (int64_t)4 * LAT * LAT);
        F(p + ".1.pwconv2.bias", LAT);
        F(p + ".1.gamma", LAT);
    }
    const int D = c.dec_dim;
    F("decoder.decoder.0.conv.weight", (int64_t)D * LAT * 7);
    F("decoder.decoder.0.conv.bias", D);
    int cin = D;
    for (int b = 0; b < 4; ++b) {
        const int r = c.dec_up_rates[b], cout = cin / 2;
        const std::string p = fmt("decoder.decoder.%d.block", b + 1);
        F(p + ".0.alpha", cin); F(p + ".0.beta", cin);
        TC(p + ".1.conv.weight", cin, cout, 2 * r, r);
        F(p + ".1.conv.bias", cout);
        for (int u = 0; u < 3; ++u) {
            const std::string q = fmt("%s.%d", p.c_str(), u + 2);
            F(q + ".act1.alpha", cout); F(q + ".act1.beta", cout);
            F(q + ".conv1.conv.weight", (int64_t)cout * cout * 7); F(q + ".conv1.conv.bias", cout);
            F(q + ".act2.alpha", cout); F(q + ".act2.beta", cout);
            F(q + ".conv2.conv.weight", (int64_t)cout * cout); F(q + ".conv2.conv.bias", cout);
        }
        cin = cout;
    }
    F("decoder.decoder.5.alpha", cin); F("decoder.decoder.5.beta", cin);
    F("decoder.decoder.6.conv.weight", (int64_t)cin * 7);
    F("decoder.decoder.6.conv.bias", 1);
}

static q3_status check_config(const q3_config& c) {
    // config.json is untrusted input (u64 values cast to int32 by the parser): every dimension positive and bounded
    // before any size is computed from it
    const struct { const char* name; int v, lo, hi; } dims[] = {
        {"text_vocab", c.text_vocab, 1, 1 << 22}, {"text_dim", c.text_dim, 32, 1 << 15}, {"hidden", c.hidden, 32, 1 << 15}, {"inter", c.inter, 32, 1 << 17},
        {"n_layers", c.n_layers, 1, 256}, {"n_heads", c.n_heads, 1, 256}, {"n_kv_heads", c.n_kv_heads, 1, 256},
        {"cp_hidden", c.cp_hidden, 32, 1 << 15}, {"cp_inter", c.cp_inter, 32, 1 << 17}, {"cp_layers", c.cp_layers, 1, 64},
        {"cp_heads", c.cp_heads, 1, 256}, {"cp_kv_heads", c.cp_kv_heads, 1, 256}, {"cp_vocab", c.cp_vocab, 2, 4096},
        {"dec_cb_dim", c.dec_cb_dim, 1, 256}, {"dec_q_dim", c.dec_q_dim, 1, 1 << 14}, {"dec_latent", c.dec_latent, 1, 1 << 14},
        {"dec_hidden", c.dec_hidden, 1, 1 << 14}, {"dec_layers", c.dec_layers, 1, 64}, {"dec_heads", c.dec_heads, 1, 256},
        {"dec_inter", c.dec_inter, 1, 1 << 16}, {"dec_cb_size", c.dec_cb_size, 2, 4096}, {"dec_dim", c.dec_dim, 16, 1 << 14},
        {"dec_up_ratios[0]", c.dec_up_ratios[0], 1, 16}, {"dec_up_ratios[1]", c.dec_up_ratios[1], 1, 16},
        {"dec_up_rates[0]", c.dec_up_rates[0], 1, 32}, {"dec_up_rates[1]", c.dec_up_rates[1], 1, 32},
        {"dec_up_rates[2]", c.dec_up_rates[2], 1, 32}, {"dec_up_rates[3]", c.dec_up_rates[3], 1, 32}};
    for (const auto& d : dims)
        if (d.v < d.lo || d.v > d.hi) return set_err(Q3_UNSUPPORTED, "config: %s = %d is outside [%d, %d]", d.name, d.v, d.lo, d.hi);
    if (c.dec_dim % 16) return set_err(Q3_UNSUPPORTED, "config: dec_dim %d must be a multiple of 16 (four halvings)", c.dec_dim);
    if (!(c.rms_eps > 0.0f) || !(c.dec_eps > 0.0f) || !(c.rope_theta > 1.0f) || !(c.dec_theta > 1.0f))
        return set_err(Q3_UNSUPPORTED, "config: eps / rope theta out of range");
    if (c.head_dim != HEAD_DIM) return set_err(Q3_UNSUPPORTED, "head_dim %d unsupported (kernels are built for 128)", c.head_dim);
    if (c.dec_head_dim != 64) return set_err(Q3_UNSUPPORTED, "decoder head_dim %d unsupported (64)", c.dec_head_dim);
    if (c.hidden % 32 || c.inter % 32 || c.text_dim % 32 || c.cp_hidden % 32 || c.cp_inter % 32)
        return set_err(Q3_UNSUPPORTED, "hidden/intermediate sizes must be multiples of 32");
    if (c.n_groups != 16) return set_err(Q3_UNSUPPORTED, "n_groups must be 16");
    if (c.codec_vocab > 4096 || c.codec_vocab < 1024) return set_err(Q3_UNSUPPORTED, "codec_vocab must be in [1024, 4096]");
    const int nrep = c.n_heads / (c.n_kv_heads ? c.n_kv_heads : 1), crep = c.cp_heads / (c.cp_kv_heads ? c.cp_kv_heads : 1);
    if ((nrep != 1 && nrep != 2 && nrep != 4) || (crep != 1 && crep != 2 && crep != 4))
        return set_err(Q3_UNSUPPORTED, "heads/kv_heads ratio must be 1, 2 or 4");
    if (c.dec_cb_dim > 256) return set_err(Q3_UNSUPPORTED, "dec_cb_dim > 256");
    return Q3_OK;
}

extern "C" q3_status q3_model_create(const q3_config* cfg, int device, q3_model** out) {
    if (!cfg || !out) return set_err(Q3_INVALID_ARG, "q3_model_create: null argument");
    Q3C(check_config(*cfg));
    if (device == -1) {   // manifest-only handle (no GPU): names/shapes for tools and CPU tests
        std::unique_ptr<q3_model> mm(new q3_model());
        mm->cfg = *cfg; mm->device = -1;
        build_manifest(mm.get());
        *out = mm.release();
        return Q3_OK;
    }
    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return set_err(Q3_INVALID_ARG, "device %d not available (%d visible)", device, ndev);
    HIPC(hipSetDevice(device));
    std::unique_ptr<q3_model> m(new q3_model());
    m->kv_pool.run_floats = (size_t)cfg->n_kv_heads * KV_PAGE_POS * HEAD_DIM; m->kv_pool.n_layers = cfg->n_layers;
    m->kv_pool16.run_floats = m->kv_pool.run_floats; m->kv_pool16.n_layers = cfg->n_layers; m->kv_pool16.elem_bytes = 2;
    m->kv_pool.budget = &m->kv_budget; m->kv_pool.unit = 2; m->kv_pool16.budget = &m->kv_budget; m->kv_pool16.unit = 1;
    m->cfg = *cfg; m->device = device;
    build_manifest(m.get());
    HIPC(hipMalloc((void**)&m->arena, m->arena_bytes));
    HIPC(hipMemset(m->arena, 0, m->arena_bytes));
    *out = m.release();
    return Q3_OK;
}

void model_destroy(q3_model* m);
extern "C" void q3_model_free(q3_model* m) {
    if (!m) return;
    if (m->refs.fetch_sub(1) == 1) model_destroy(m);       // the handle's reference; the last session's ~q3_session drops the last one otherwise
}
void model_destroy(q3_model* m) {
    if (m->device < 0) { delete m; return; }
    hipSetDevice(m->device);
    hipFree(m->arena); hipFree(m->rope_cos); hipFree(m->rope_sin); hipFree(m->derived); hipFree(m->wpk_arena);
    hipFree((void*)m->rest_cbs_dev); hipFree((void*)m->cp_embs_dev); hipFree(m->proj_tabs); hipFree(m->qkv0_tabs);
    for (hipStream_t st : m->idle_streams) (void)hipStreamDestroy(st);
    delete m;
}

// Paged KV pool of the model (KvPool above). limit: the most pages sessions may hold at once (0 = HBM is the limit); a session
// that needs a page beyond it fails with Q3_KV_OVERFLOW — the reference's KV-overflow bail (kv_cache.rs:293-300).
extern "C" q3_status q3_model_set_codec_planes(q3_model* m, int planes) {
    if (!m) return set_err(Q3_INVALID_ARG, "q3_model_set_codec_planes: null model");
    if (planes != 2 && planes != 3) return set_err(Q3_INVALID_ARG, "q3_model_set_codec_planes: %d (2 or 3)", planes);
    m->codec_planes = planes;
    return Q3_OK;
}

extern "C" q3_status q3_model_kv_pool_limit(q3_model* m, int max_pages) {
    if (!m || m->device < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_limit: no device model");
    if (max_pages < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_limit: negative limit");
    std::lock_guard<std::mutex> g(m->kv_budget.mu);
    if (max_pages > 0 && 2L * max_pages < m->kv_budget.used)
        return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_limit: %ld pages (f32 equivalents) are in use", (m->kv_budget.used + 1) / 2);
    m->kv_budget.limit = 2L * max_pages;
    return Q3_OK;
}
// Pages are counted in f32 equivalents: a page of a bf16 session (q3_session_set_kv_dtype) is half of one, rounded up in the
// totals below — one budget covers both pools.
extern "C" q3_status q3_model_kv_pool_info(q3_model* m, int* page_positions, size_t* page_bytes, int* pages_total, int* pages_in_use, int* pages_peak) {
    if (!m || m->device < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_info: no device model");
    int tot = 0;
    { std::lock_guard<std::mutex> g(m->kv_pool.mu); tot += m->kv_pool.total; }
    { std::lock_guard<std::mutex> g(m->kv_pool16.mu); tot += (m->kv_pool16.total + 1) / 2; }
    std::lock_guard<std::mutex> g(m->kv_budget.mu);
    if (page_positions) *page_positions = KV_PAGE_POS;
    if (page_bytes) *page_bytes = m->kv_pool.page_bytes();
    if (pages_total) *pages_total = tot;
    if (pages_in_use) *pages_in_use = (int)((m->kv_budget.used + 1) / 2);
    if (pages_peak) *pages_peak = (int)((m->kv_budget.peak + 1) / 2);
    return Q3_OK;
}
// Slabs of either pool none of whose pages is held go back to the device (a server that has seen one very long prompt need not
// keep its ~1 GB slabs for the model's lifetime). Safe beside running sessions: held pages pin their slab.
extern "C" q3_status q3_model_kv_pool_trim(q3_model* m, size_t* bytes_freed) {
    if (!m || m->device < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_trim: no device model");
    HIPC(hipSetDevice(m->device));
    const size_t n = m->kv_pool.trim() + m->kv_pool16.trim();
    if (bytes_freed) *bytes_freed = n;
    return Q3_OK;
}

extern "C" q3_status q3_model_config(const q3_model* m, q3_config* out) {
    if (!m || !out) return set_err(Q3_INVALID_ARG, "q3_model_config: null argument");
    *out = m->cfg;
    return Q3_OK;
}
extern "C" int q3_model_n_tensors(const q3_model* m) { return m ? (int)m->slots.size() : 0; }
extern "C" q3_status q3_model_tensor_info(const q3_model* m, int i, const char** name, int64_t* n, int* stored_dtype) {
    if (!m || i < 0 || i >= (int)m->slots.size()) return set_err(Q3_INVALID_ARG, "tensor index out of range");
    if (name) *name = m->slots[i].name.c_str();
    if (n) *n = m->slots[i].n;
    if (stored_dtype) *stored_dtype = m->slots[i].stored;
    return Q3_OK;
}

// row-major [N][K] bf16 → MFMA tiles, zero padded.
//   mode 1: [N↑16/16][K↑32/32][lane][8],  lane = (k-group << 4) | row      (16 rows x 32 k per KiB)
//   mode 2: [N↑4/4][K↑128/128][lane][8],  lane = (k-group << 2) | row      (4 rows x 128 k per KiB)
void retile_bf16(const uint16_t* src, int N, int K, uint16_t* dst, int mode) {
    const int RT = mode == 2 ? 4 : 16, KS = mode == 2 ? 128 : 32, RB = mode == 2 ? 2 : 4;
    const int T = (N + RT - 1) / RT, S = (K + KS - 1) / KS;
    auto body = [=](int t0, int t1) {
        for (int t = t0; t < t1; ++t)
            for (int s = 0; s < S; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int n = t * RT + (l & (RT - 1)), k0 = s * KS + (l >> RB) * 8;
                    uint16_t* d = dst + (((size_t)t * S + s) * 64 + l) * 8;
                    for (int e = 0; e < 8; ++e) d[e] = (n < N && k0 + e < K) ? src[(size_t)n * K + k0 + e] : (uint16_t)0;
                }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 2 || (size_t)N * K < (1u << 20)) { body(0, T); return; }
    std::vector<std::thread> th;
    const int per = (T + (int)nt - 1) / (int)nt;
    for (unsigned i = 0; i < nt; ++i) {
        const int a = (int)i * per, b = a + per < T ? a + per : T;
        if (a < b) th.emplace_back(body, a, b);
    }
    for (auto& x : th) x.join();
}

extern "C" q3_status q3_model_set_tensor(q3_model* m, const char* name, int dtype, const void* data, int64_t n) {
    if (!m || !name || !data) return set_err(Q3_INVALID_ARG, "q3_model_set_tensor: null argument");
    auto it = m->index.find(name);
    if (it == m->index.end()) return set_err(Q3_INVALID_ARG, "unknown tensor name: %s", name);
    Slot& s = m->slots[it->second];
    if (n != s.n) return set_err(Q3_INVALID_ARG, "tensor %s has %lld elements, expected %lld", name, (long long)n, (long long)s.n);
    if (m->device < 0) return set_err(Q3_UNSUPPORTED, "manifest-only model handle (device -1) holds no weights");
    HIPC(hipSetDevice(m->device));
    const size_t bytes = (size_t)n * (s.stored == Q3_DTYPE_BF16 ? 2 : 4);
    std::vector<char> tmp;
    const void* src = data;
    size_t up_bytes = bytes;
    if (s.kind == SK_TILED) {
        std::vector<uint16_t> w((size_t)n);
        if (dtype == Q3_DTYPE_BF16) memcpy(w.data(), data, (size_t)n * 2);
        else if (dtype == Q3_DTYPE_F32) for (int64_t i = 0; i < n; ++i) w[(size_t)i] = f32_to_bf16_host(((const float*)data)[i]);
        else return set_err(Q3_INVALID_ARG, "unsupported source dtype %d", dtype);
        up_bytes = tiled_elems(1, s.rows, s.cols) * 2;
        tmp.resize(up_bytes);
        retile_bf16(w.data(), s.rows, s.cols, (uint16_t*)tmp.data(), 1);
        src = tmp.data();
        if (s.dual) {
            std::vector<uint16_t> t2(tiled_elems(2, s.rows, s.cols));
            retile_bf16(w.data(), s.rows, s.cols, t2.data(), 2);
            HIPC(q3_hipMemcpy(m->arena + s.offset2, t2.data(), t2.size() * 2, hipMemcpyHostToDevice));
        }
    } else if (s.kind == SK_TRANSCONV) {
        // [cin][cout][k] → per-phase causal-conv weights [stride][cout][cin][taps]
        std::vector<float> w((size_t)n);
        if (dtype == Q3_DTYPE_F32) memcpy(w.data(), data, (size_t)n * 4);
        else for (int64_t i = 0; i < n; ++i) w[(size_t)i] = bf16_to_f32_host(((const uint16_t*)data)[i]);
        const int cin = s.tc_cin, cout = s.tc_cout, k = s.tc_k, st = s.tc_stride, taps = k / st;
        tmp.resize(bytes);
        float* o = (float*)tmp.data();
        for (int ph = 0; ph < st; ++ph)
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int tp = 0; tp < taps; ++tp) {
                        // tap tp multiplies x[j - (taps-1-tp)]  ⇒  kernel index ph + (taps-1-tp)*stride
                        const int kk = ph + (taps - 1 - tp) * st;
                        o[(((size_t)ph * cout + co) * cin + ci) * taps + tp] = w[((size_t)ci * cout + co) * k + kk];
                    }
        src = tmp.data();
    } else if (s.stored == Q3_DTYPE_BF16 && dtype == Q3_DTYPE_F32) {
        tmp.resize(bytes);
        uint16_t* o = (uint16_t*)tmp.data(); const float* f = (const float*)data;
        for (int64_t i = 0; i < n; ++i) o[i] = f32_to_bf16_host(f[i]);
        src = tmp.data();
    } else if (s.stored == Q3_DTYPE_F32 && dtype == Q3_DTYPE_BF16) {
        tmp.resize(bytes);
        float* o = (float*)tmp.data(); const uint16_t* h = (const uint16_t*)data;
        for (int64_t i = 0; i < n; ++i) o[i] = bf16_to_f32_host(h[i]);
        src = tmp.data();
    } else if (dtype != Q3_DTYPE_F32 && dtype != Q3_DTYPE_BF16) {
        return set_err(Q3_INVALID_ARG, "unsupported source dtype %d", dtype);
    }
    HIPC(q3_hipMemcpy(m->arena + s.offset, src, up_bytes, hipMemcpyHostToDevice));
    s.loaded = true;
    m->finalized = false;
    return Q3_OK;
}

extern "C" q3_status q3_model_arena(q3_model* m, void** dev_ptr, size_t* bytes) {
    if (!m) return set_err(Q3_INVALID_ARG, "null model");
    if (dev_ptr) *dev_ptr = m->arena;
    if (bytes) *bytes = m->arena_bytes;
    return Q3_OK;
}
extern "C" q3_status q3_model_mark_loaded(q3_model* m) {
    if (!m) return set_err(Q3_INVALID_ARG, "null model");
    for (auto& s : m->slots) s.loaded = true;
    return Q3_OK;
}

template <typename T>
static const T* P(const q3_model* m, const std::string& name) {
    auto it = m->index.find(name);
    if (it == m->index.end()) return nullptr;
    return (const T*)(m->arena + m->slots[it->second].offset);
}
static TW PT(const q3_model* m, const std::string& name) {
    TW w;
    auto it = m->index.find(name);
    if (it == m->index.end()) return w;
    const Slot& s = m->slots[it->second];
    w.t1 = (const uint16_t*)(m->arena + s.offset);
    if (s.dual) w.t2 = (const uint16_t*)(m->arena + s.offset2);
    return w;
}
static void resolve_layer(const q3_model* m, LayerW& L, const std::string& p) {
    L.in_ln = P<float>(m, p + ".input_layernorm.weight");
    L.qkv = PT(m, p + ".self_attn.q_proj.weight");
    L.o = PT(m, p + ".self_attn.o_proj.weight");
    L.q_norm = P<float>(m, p + ".self_attn.q_norm.weight");
    L.k_norm = P<float>(m, p + ".self_attn.k_norm.weight");
    L.post_ln = P<float>(m, p + ".post_attention_layernorm.weight");
    L.gate = PT(m, p + ".mlp.gate_proj.weight");
    L.up = PT(m, p + ".mlp.up_proj.weight");
    L.down = PT(m, p + ".mlp.down_proj.weight");
}

extern "C" q3_status q3_model_finalize(q3_model* m) {
    if (!m) return set_err(Q3_INVALID_ARG, "null model");
    if (m->device < 0) return set_err(Q3_UNSUPPORTED, "manifest-only model handle (device -1) cannot be finalized");
    for (auto& s : m->slots)
        if (!s.loaded) return set_err(Q3_MISSING_WEIGHT, "Missing weight: %s", s.name.c_str());
    HIPC(hipSetDevice(m->device));
    const q3_config& c = m->cfg;
    m->text_emb = P<uint16_t>(m, "talker.model.text_embedding.weight");
    m->fc1w = PT(m, "talker.text_projection.linear_fc1.weight");
    m->fc1b = P<float>(m, "talker.text_projection.linear_fc1.bias");
    m->fc2w = PT(m, "talker.text_projection.linear_fc2.weight");
    m->fc2b = P<float>(m, "talker.text_projection.linear_fc2.bias");
    m->codec_emb = P<uint16_t>(m, "talker.model.codec_embedding.weight");
    m->norm = P<float>(m, "talker.model.norm.weight");
    m->codec_head = PT(m, "talker.codec_head.weight");
    m->mtp_w = PT(m, "talker.code_predictor.small_to_mtp_projection.weight");
    m->mtp_b = P<float>(m, "talker.code_predictor.small_to_mtp_projection.bias");
    m->cp_norm = P<float>(m, "talker.code_predictor.model.norm.weight");
    m->tl.resize(c.n_layers); m->cl.resize(c.cp_layers);
    for (int i = 0; i < c.n_layers; ++i) resolve_layer(m, m->tl[i], fmt("talker.model.layers.%d", i));
    for (int i = 0; i < c.cp_layers; ++i) resolve_layer(m, m->cl[i], fmt("talker.code_predictor.model.layers.%d", i));
    m->cp_emb.resize(15); m->cp_head.resize(15);
    for (int g = 0; g < 15; ++g) {
        m->cp_emb[g] = P<uint16_t>(m, fmt("talker.code_predictor.model.codec_embedding.%d.weight", g));
        m->cp_head[g] = PT(m, fmt("talker.code_predictor.lm_head.%d.weight", g));
    }
    if (!m->cp_embs_dev) HIPC(hipMalloc((void**)&m->cp_embs_dev, 15 * sizeof(void*)));
    HIPC(q3_hipMemcpy((void*)m->cp_embs_dev, m->cp_emb.data(), 15 * sizeof(void*), hipMemcpyHostToDevice));
    if (m->mtp_w.t1) {
        // Pre-projected embedding tables (1.7B): 15 x [cp_vocab][CH] + [codec_vocab][CH] f32 (138 MB). Built with the very
        // GEMV launches the frame loop would use (8 gathered rows per launch), so a table row is exactly what the
        // per-pass projection of that row computes at a batch of 8.
        const int Hh = c.hidden, CHh = c.cp_hidden;
        const size_t total = ((size_t)15 * c.cp_vocab + c.codec_vocab) * CHh;
        if (!m->proj_tabs) HIPC(hipMalloc((void**)&m->proj_tabs, total * 4));
        float* xin = nullptr; uint32_t* ids = nullptr;
        HIPC(hipMalloc((void**)&xin, (size_t)8 * Hh * 4)); HIPC(hipMalloc((void**)&ids, 8 * 4));
        float* cur = m->proj_tabs;
        hipError_t e = hipSuccess;
        for (int tbl = 0; tbl <= 15 && e == hipSuccess; ++tbl) {
            const uint16_t* emb = tbl < 15 ? m->cp_emb[tbl] : m->codec_emb;
            const int rows = tbl < 15 ? c.cp_vocab : c.codec_vocab;
            if (tbl < 15) m->cp_proj[tbl] = cur; else m->sem_proj = cur;
            for (int r0 = 0; r0 < rows && e == hipSuccess; r0 += 8) {
                const int M = (rows - r0) < 8 ? (rows - r0) : 8;
                uint32_t h[8]; for (int i = 0; i < 8; ++i) h[i] = (uint32_t)(r0 + (i < M ? i : 0));
                e = hipMemcpyAsync(ids, h, sizeof h, hipMemcpyHostToDevice, 0);
                if (e == hipSuccess) e = launch_gather_rows_bf16(emb, ids, xin, M, Hh, 0);
                LinArgs a;
                a.N = CHh; a.K = Hh; set_w(a, m->mtp_w, 8, CHh, Hh); a.x = xin; a.ldx = Hh; a.bias = m->mtp_b; a.y = cur + (size_t)r0 * CHh; a.ldy = CHh;
                a.M = M; a.epi = EPI_NONE;
                if (e == hipSuccess) e = launch_linear(a, 0);
                if (e == hipSuccess) e = hipStreamSynchronize(0);      // `h` is reused by the next iteration
            }
            cur += (size_t)rows * CHh;
        }
        hipFree(xin); hipFree(ids);
        if (e != hipSuccess) return set_err(Q3_HIP_ERROR, "projection tables: %s", hipGetErrorString(e));
    }
    if (c.cp_hidden % 4 == 0 && (m->mtp_w.t1 ? m->proj_tabs != nullptr : c.cp_hidden == c.hidden)) {
        // Layer-0 q|k|v tables: the first code-predictor layer sees only (projected) embedding rows on passes >= 1, so
        // RMSNorm + qkv of every possible row is computed once, again with the launches the frame loop would use.
        const int CHh = c.cp_hidden, QKVD = (c.cp_heads + 2 * c.cp_kv_heads) * HEAD_DIM;
        const size_t total = ((size_t)15 * c.cp_vocab + c.codec_vocab) * QKVD;
        if (!m->qkv0_tabs) HIPC(hipMalloc((void**)&m->qkv0_tabs, total * 4));
        float* xin = nullptr; uint32_t* ids = nullptr;
        HIPC(hipMalloc((void**)&xin, (size_t)8 * CHh * 4)); HIPC(hipMalloc((void**)&ids, 8 * 4));
        float* cur = m->qkv0_tabs;
        hipError_t e = hipSuccess;
        for (int tbl = 0; tbl <= 15 && e == hipSuccess; ++tbl) {
            const int rows = tbl < 15 ? c.cp_vocab : c.codec_vocab;
            const float* prow = m->mtp_w.t1 ? (tbl < 15 ? m->cp_proj[tbl] : m->sem_proj) : nullptr;
            const uint16_t* emb = tbl < 15 ? m->cp_emb[tbl] : m->codec_emb;
            if (tbl < 15) m->cp_qkv0[tbl] = cur; else m->sem_qkv0 = cur;
            for (int r0 = 0; r0 < rows && e == hipSuccess; r0 += 8) {
                const int M = (rows - r0) < 8 ? (rows - r0) : 8;
                const float* x = prow ? prow + (size_t)r0 * CHh : xin;
                if (!prow) {
                    uint32_t h[8]; for (int i = 0; i < 8; ++i) h[i] = (uint32_t)(r0 + (i < M ? i : 0));
                    e = hipMemcpyAsync(ids, h, sizeof h, hipMemcpyHostToDevice, 0);
                    if (e == hipSuccess) e = launch_gather_rows_bf16(emb, ids, xin, M, CHh, 0);
                    if (e == hipSuccess) e = hipStreamSynchronize(0);
                }
                LinArgs a;
                a.N = QKVD; a.K = CHh; set_w(a, m->cl[0].qkv, 8, QKVD, CHh); a.x = x; a.ldx = CHh; a.norm_w = m->cl[0].in_ln; a.eps = c.rms_eps;
                a.y = cur + (size_t)r0 * QKVD; a.ldy = QKVD; a.M = M; a.epi = EPI_NONE;
                if (e == hipSuccess) e = launch_linear(a, 0);
                if (!prow && e == hipSuccess) e = hipStreamSynchronize(0);
            }
            cur += (size_t)rows * QKVD;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(0);
        hipFree(xin); hipFree(ids);
        if (e != hipSuccess) return set_err(Q3_HIP_ERROR, "layer-0 qkv tables: %s", hipGetErrorString(e));
    }

    // RoPE tables on the host with libm (bit-identical to the CPU oracle): transformer.rs:78-92, 133-175
    if (!m->rope_cos) {
        m->rope_len = 8192;
        std::vector<float> cs((size_t)m->rope_len * 64), sn((size_t)m->rope_len * 64);
        for (int i = 0; i < 64; ++i) {
            const float inv = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)HEAD_DIM);
            for (int p = 0; p < m->rope_len; ++p) {
                const float f = (float)p * inv;
                cs[(size_t)p * 64 + i] = cosf(f); sn[(size_t)p * 64 + i] = sinf(f);
            }
        }
        HIPC(hipMalloc((void**)&m->rope_cos, cs.size() * 4));
        HIPC(hipMalloc((void**)&m->rope_sin, sn.size() * 4));
        HIPC(q3_hipMemcpy(m->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
        HIPC(q3_hipMemcpy(m->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    }

    // decoder pointers + derived tensors (normalised codebooks, snake tables)
    const int CB = c.dec_cb_size, CD = c.dec_cb_dim;
    size_t n_snake = 0;
    { int cin = c.dec_dim; for (int b = 0; b < 4; ++b) { n_snake += cin; cin /= 2; n_snake += (size_t)cin * 6; } n_snake += cin; }
    const size_t derived_floats = (size_t)16 * CB * CD + 2 * n_snake;
    if (!m->derived) HIPC(hipMalloc((void**)&m->derived, derived_floats * 4));
    float* cursor = m->derived;
    std::vector<const float*> rest(15);
    auto CBOOK = [&](const std::string& p) -> const float* {
        float* dst = cursor; cursor += (size_t)CB * CD;
        launch_norm_codebook(P<float>(m, p + "._codebook.embedding_sum"), P<float>(m, p + "._codebook.cluster_usage"), dst, CB, CD, 0);
        return dst;
    };
    m->first_cb = CBOOK("decoder.quantizer.rvq_first.vq.layers.0");
    for (int i = 0; i < 15; ++i) rest[i] = CBOOK(fmt("decoder.quantizer.rvq_rest.vq.layers.%d", i));
    if (!m->rest_cbs_dev) HIPC(hipMalloc((void**)&m->rest_cbs_dev, 15 * sizeof(void*)));
    HIPC(q3_hipMemcpy((void*)m->rest_cbs_dev, rest.data(), 15 * sizeof(void*), hipMemcpyHostToDevice));
    auto SNAKE = [&](const std::string& pa, const std::string& pb, int C, const float*& a, const float*& ib) {
        float* da = cursor; cursor += C; float* di = cursor; cursor += C;
        launch_snake_tables(P<float>(m, pa), P<float>(m, pb), da, di, C, 0);
        a = da; ib = di;
    };
    m->first_proj = P<float>(m, "decoder.quantizer.rvq_first.output_proj.weight");
    m->rest_proj = P<float>(m, "decoder.quantizer.rvq_rest.output_proj.weight");
    m->pre_w = P<float>(m, "decoder.pre_conv.conv.weight"); m->pre_b = P<float>(m, "decoder.pre_conv.conv.bias");
    m->inp_w = P<float>(m, "decoder.pre_transformer.input_proj.weight"); m->inp_b = P<float>(m, "decoder.pre_transformer.input_proj.bias");
    m->outp_w = P<float>(m, "decoder.pre_transformer.output_proj.weight"); m->outp_b = P<float>(m, "decoder.pre_transformer.output_proj.bias");
    m->dec_norm = P<float>(m, "decoder.pre_transformer.norm.weight");
    m->dl.resize(c.dec_layers);
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = fmt("decoder.pre_transformer.layers.%d", i);
        DecLayerW& L = m->dl[i];
        L.in_ln = P<float>(m, p + ".input_layernorm.weight");
        L.q = P<float>(m, p + ".self_attn.q_proj.weight"); L.k = P<float>(m, p + ".self_attn.k_proj.weight");
        L.v = P<float>(m, p + ".self_attn.v_proj.weight"); L.o = P<float>(m, p + ".self_attn.o_proj.weight");
        L.attn_scale = P<float>(m, p + ".self_attn_layer_scale.scale");
        L.post_ln = P<float>(m, p + ".post_attention_layernorm.weight");
        L.gate = P<float>(m, p + ".mlp.gate_proj.weight"); L.up = P<float>(m, p + ".mlp.up_proj.weight");
        L.down = P<float>(m, p + ".mlp.down_proj.weight"); L.mlp_scale = P<float>(m, p + ".mlp_layer_scale.scale");
    }
    for (int i = 0; i < 2; ++i) {
        const std::string p = fmt("decoder.upsample.%d", i);
        UpW& U = m->up[i]; U.ratio = c.dec_up_ratios[i];
        U.tw = P<float>(m, p + ".0.conv.weight"); U.tb = P<float>(m, p + ".0.conv.bias");
        U.dww = P<float>(m, p + ".1.dwconv.conv.weight"); U.dwb = P<float>(m, p + ".1.dwconv.conv.bias");
        U.nw = P<float>(m, p + ".1.norm.weight"); U.nb = P<float>(m, p + ".1.norm.bias");
        U.p1w = P<float>(m, p + ".1.pwconv1.weight"); U.p1b = P<float>(m, p + ".1.pwconv1.bias");
        U.p2w = P<float>(m, p + ".1.pwconv2.weight"); U.p2b = P<float>(m, p + ".1.pwconv2.bias");
        U.gamma = P<float>(m, p + ".1.gamma");
    }
    m->init_w = P<float>(m, "decoder.decoder.0.conv.weight"); m->init_b = P<float>(m, "decoder.decoder.0.conv.bias");
    int cin = c.dec_dim;
    for (int b = 0; b < 4; ++b) {
        DecBlockW& B = m->blk[b];
        const std::string p = fmt("decoder.decoder.%d.block", b + 1);
        B.cin = cin; B.cout = cin / 2; B.rate = c.dec_up_rates[b];
        SNAKE(p + ".0.alpha", p + ".0.beta", cin, B.a, B.ib);
        B.tw = P<float>(m, p + ".1.conv.weight"); B.tb = P<float>(m, p + ".1.conv.bias");
        for (int u = 0; u < 3; ++u) {
            const std::string q = fmt("%s.%d", p.c_str(), u + 2);
            ResUnitW& R = B.res[u];
            SNAKE(q + ".act1.alpha", q + ".act1.beta", B.cout, R.a1, R.ib1);
            R.c1w = P<float>(m, q + ".conv1.conv.weight"); R.c1b = P<float>(m, q + ".conv1.conv.bias");
            SNAKE(q + ".act2.alpha", q + ".act2.beta", B.cout, R.a2, R.ib2);
            R.c2w = P<float>(m, q + ".conv2.conv.weight"); R.c2b = P<float>(m, q + ".conv2.conv.bias");
        }
        cin = B.cout;
    }
    SNAKE("decoder.decoder.5.alpha", "decoder.decoder.5.beta", cin, m->fin_a, m->fin_ib);
    m->fin_w = P<float>(m, "decoder.decoder.6.conv.weight"); m->fin_b = P<float>(m, "decoder.decoder.6.conv.bias");
    // bf16x3 copies of every vocoder conv / linear weight the matrix-core kernel can take (cout % 32 == 0, cin % 16 == 0):
    // (pointer, cout, cin, taps, phases); transposed convs are already stored per phase [stride][cout][cin][taps]
    {
        struct PW { const float* w; int cout, cin, k, phases; };
        std::vector<PW> list;
        const int CDm = c.dec_cb_dim, Qm = c.dec_q_dim, LATm = c.dec_latent, DHm = c.dec_hidden, QDm = c.dec_heads * c.dec_head_dim, DIm = c.dec_inter;
        list.push_back({m->first_proj, Qm, CDm, 1, 1}); list.push_back({m->rest_proj, Qm, CDm, 1, 1});
        list.push_back({m->pre_w, LATm, Qm, 3, 1});
        list.push_back({m->inp_w, DHm, LATm, 1, 1}); list.push_back({m->outp_w, LATm, DHm, 1, 1});
        for (auto& L : m->dl) {
            list.push_back({L.q, QDm, DHm, 1, 1}); list.push_back({L.k, QDm, DHm, 1, 1}); list.push_back({L.v, QDm, DHm, 1, 1});
            list.push_back({L.o, DHm, QDm, 1, 1}); list.push_back({L.gate, DIm, DHm, 1, 1}); list.push_back({L.up, DIm, DHm, 1, 1});
            list.push_back({L.down, DHm, DIm, 1, 1});
        }
        for (int i = 0; i < 2; ++i) {
            list.push_back({m->up[i].tw, LATm, LATm, 1, m->up[i].ratio});
            list.push_back({m->up[i].p1w, 4 * LATm, LATm, 1, 1}); list.push_back({m->up[i].p2w, LATm, 4 * LATm, 1, 1});
        }
        list.push_back({m->init_w, c.dec_dim, LATm, 7, 1});
        for (int b = 0; b < 4; ++b) {
            const DecBlockW& B = m->blk[b];
            list.push_back({B.tw, B.cout, B.cin, 2, B.rate});
            for (int u = 0; u < 3; ++u) { list.push_back({B.res[u].c1w, B.cout, B.cout, 7, 1}); list.push_back({B.res[u].c2w, B.cout, B.cout, 1, 1}); }
        }
        size_t total = 0;
        for (auto& e : list) if (e.cout % 32 == 0 && e.cin % 16 == 0) total += packed_conv_w_bytes(e.cout, e.cin, e.k) * (size_t)e.phases;
        m->wpk.clear();
        if (total) {
            if (!m->wpk_arena) HIPC(hipMalloc(&m->wpk_arena, total));
            char* cur = (char*)m->wpk_arena;
            for (auto& e : list) {
                if (e.cout % 32 || e.cin % 16) continue;
                const size_t per = packed_conv_w_bytes(e.cout, e.cin, e.k);
                m->wpk[e.w] = cur;
                for (int ph = 0; ph < e.phases; ++ph)
                    HIPC(launch_pack_conv_w(e.w + (size_t)ph * e.cout * e.cin * e.k, cur + (size_t)ph * per, e.cout, e.cin, e.k, 0));
                cur += per * (size_t)e.phases;
            }
        }
    }
    HIPC(hipGetLastError());
    HIPC(q3_hipDeviceSynchronize());
    // the f32 page pool's first slab now, not inside the first session's prefill (its ~1 GB hipMalloc sat on the first request's
    // time to first audio); a failure here is not fatal — the first session will report it. Q3_KV_NO_PREWARM=1: lazily, as before
    if (!getenv("Q3_KV_NO_PREWARM")) { if (m->kv_pool.prewarm() != hipSuccess) (void)hipGetLastError(); }
    m->finalized = true;
    return Q3_OK;
}

