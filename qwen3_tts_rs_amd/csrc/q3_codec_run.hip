// q3_codec_run.hip — the device-memory cache, the vocoder pipeline (codec_decode_dev) and q3_decode_codes
// (one of the five units of the engine: q3_engine.h says which holds what)
#include "q3_engine.h"

// ------------------------------------------------------------------------------------------------
// device buffer helpers
// ------------------------------------------------------------------------------------------------
// Session-lifetime buffers (KV pages, workspaces, vocoder scratch) come from a small per-device cache of exact-size
// blocks: a server creates sessions of a few recurring shapes, and hipMalloc / hipFree of multi-GB blocks on the
// request path costs driver time that varies from box to box (VRAM clearing, page-table work) — up to 150 ms per
// 16-utterance session was seen in otherwise identical runs. Blocks are returned only by owners that have
// synchronised the streams that used them (q3_session_free, prefill_gemm); at most Q3_DEV_CACHE_MB (default 32768)
// MB stay cached per process, beyond that blocks go back to the driver. HBM is 288 GB: capacity is not the constraint.
namespace {
struct DevCache {
    std::mutex mu;
    std::unordered_multimap<uint64_t, void*> free_blocks;      // key = device << 48 | bytes
    std::unordered_map<void*, uint64_t> live;                  // blocks handed out by get()
    size_t cached = 0, cap = 0;
    DevCache() { const char* e = getenv("Q3_DEV_CACHE_MB"); cap = (size_t)(e ? atol(e) : 32768) << 20; }
    static uint64_t key(int dev, size_t bytes) { return ((uint64_t)dev << 48) | (uint64_t)bytes; }
    // size classes: 16 per octave (<= 6.25 % slack), so that the side sessions of a continuous-batching server — one KV extent
    // per prompt length — reuse each other's blocks instead of leaving one cached block per distinct length
    static size_t size_class(size_t bytes) {
        if (bytes <= 4096) return (bytes + 255) & ~(size_t)255;
        size_t p2 = 1; while ((p2 << 1) <= bytes) p2 <<= 1;
        const size_t step = p2 >> 4;
        return (bytes + step - 1) / step * step;
    }
    hipError_t get(void** p, size_t bytes) {
        bytes = size_class(bytes);
        int dev = 0; (void)hipGetDevice(&dev);
        const uint64_t k = key(dev, bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = free_blocks.find(k);
            if (it != free_blocks.end()) { *p = it->second; free_blocks.erase(it); cached -= bytes; live[*p] = k; return hipSuccess; }
        }
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) {                                  // out of memory: give the cache back and retry once
            trim(0);
            e = hipMalloc(p, bytes);
            if (e != hipSuccess) return e;
        }
        std::lock_guard<std::mutex> g(mu);
        live[*p] = k;
        return hipSuccess;
    }
    void put(void* p) {
        if (!p) return;
        uint64_t k = 0; bool known = false;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live.find(p);
            if (it != live.end()) { k = it->second; known = true; live.erase(it); }
            const size_t bytes = (size_t)(k & 0xffffffffffffull);
            if (known && cap && cached + bytes <= cap) { free_blocks.emplace(k, p); cached += bytes; return; }
        }
        (void)hipFree(p);
    }
    void trim(size_t keep) {
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto it = free_blocks.begin(); it != free_blocks.end() && cached > keep;) {
                cached -= (size_t)(it->first & 0xffffffffffffull); drop.push_back(it->second); it = free_blocks.erase(it);
            }
        }
        for (void* p : drop) (void)hipFree(p);
    }
};
DevCache& dev_cache() { static DevCache* c = new DevCache(); return *c; }    // leaked on purpose: outlives every static destructor
}  // namespace
hipError_t dev_malloc(void** p, size_t bytes) { return dev_cache().get(p, bytes ? bytes : 4); }
void dev_free(void* p) { dev_cache().put(p); }

// T = frames through the convolutional stack in one call, Tf = frames through the front (Tf >= T)
q3_status codec_reserve(const q3_model* m, CodecWS& ws, int T, int Tf) {
    if (Tf < T) Tf = T;
    if (T <= ws.cap_frames && Tf <= ws.cap_front) return Q3_OK;
    if (T < ws.cap_frames) T = ws.cap_frames;
    if (Tf < ws.cap_front) Tf = ws.cap_front;
    if (ws.bufA) HIPC(q3_hipDeviceSynchronize());        // growing: the old blocks go back to the cache, nothing may still be using them
    ws.release();
    const q3_config& c = m->cfg;
    int up = 1; for (int i = 0; i < 2; ++i) up *= c.dec_up_ratios[i];
    // largest [C][L] activation per frame
    size_t per = (size_t)4 * c.dec_latent * up;                 // ConvNeXt hidden 4*LAT × (T*up)
    { size_t L = up; int C = c.dec_dim; per = per > (size_t)C * L ? per : (size_t)C * L;
      for (int b = 0; b < 4; ++b) { L *= c.dec_up_rates[b]; C /= 2; if ((size_t)C * L > per) per = (size_t)C * L; } }
    const int QDm = c.dec_heads * c.dec_head_dim;
    // front: A holds q|k|v|attn-out (4*QD rows), B gate|up (2*DI) or the quantiser output, C the latent — per front frame
    size_t per_front = (size_t)4 * QDm;
    if ((size_t)2 * c.dec_inter > per_front) per_front = (size_t)2 * c.dec_inter;
    if ((size_t)2 * c.dec_cb_dim > per_front) per_front = (size_t)2 * c.dec_cb_dim;
    if ((size_t)c.dec_latent > per_front) per_front = (size_t)c.dec_latent;
    if ((size_t)c.dec_q_dim > per_front) per_front = (size_t)c.dec_q_dim;
    size_t n = per * (size_t)T;
    if (per_front * (size_t)Tf > n) n = per_front * (size_t)Tf;
    HIPC(dev_malloc((void**)&ws.bufA, n * 4)); HIPC(dev_malloc((void**)&ws.bufB, n * 4)); HIPC(dev_malloc((void**)&ws.bufC, n * 4));
    HIPC(dev_malloc((void**)&ws.bufF, n * 4));
    const size_t small = (size_t)Tf * (size_t)(QDm > c.dec_latent ? QDm : c.dec_latent);
    HIPC(dev_malloc((void**)&ws.bufD, small * 4)); HIPC(dev_malloc((void**)&ws.bufE, small * 4));
    HIPC(dev_malloc((void**)&ws.cs, (size_t)Tf * 32 * 4)); HIPC(dev_malloc((void**)&ws.sn, (size_t)Tf * 32 * 4));
    HIPC(dev_malloc((void**)&ws.frames, (size_t)Tf * 16 * 4));
    size_t total_up = up; for (int b = 0; b < 4; ++b) total_up *= c.dec_up_rates[b];
    HIPC(dev_malloc((void**)&ws.pcm, (size_t)T * total_up * 4));
    // RoPE table of the pre-transformer (decoder_12hz.rs:541-553), host libm
    std::vector<float> cs((size_t)Tf * 32), sn((size_t)Tf * 32);
    for (int i = 0; i < 32; ++i) {
        const float inv = 1.0f / powf(c.dec_theta, (float)(2 * i) / (float)c.dec_head_dim);
        for (int t = 0; t < Tf; ++t) { const float f = (float)t * inv; cs[(size_t)t * 32 + i] = cosf(f); sn[(size_t)t * 32 + i] = sinf(f); }
    }
    HIPC(q3_hipMemcpy(ws.cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    HIPC(q3_hipMemcpy(ws.sn, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    ws.cap_frames = T; ws.cap_front = Tf;
    return Q3_OK;
}

int samples_per_frame(const q3_config& c) {
    int u = 1; for (int i = 0; i < 2; ++i) u *= c.dec_up_ratios[i]; for (int i = 0; i < 4; ++i) u *= c.dec_up_rates[i];
    return u;
}

static thread_local const q3_model* tl_codec_model = nullptr;     // set by codec_decode_dev: packed-weight lookup of the helpers below
static const void* packed_of(const float* w) { return tl_codec_model ? tl_codec_model->pk(w) : nullptr; }
// bf16 planes per operand in the vocoder's matrix-core convs (q3_model_set_codec_planes): only codec_decode_dev sets 2,
// for its own launches — the encoders that share the kernels (speaker / speech tokenizer) always run the exact products
static thread_local int tl_codec_planes = 3;
struct CodecPlanesScope { explicit CodecPlanesScope(int p) { tl_codec_planes = p; } ~CodecPlanesScope() { tl_codec_planes = 3; } };
static hipError_t conv1(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L, hipStream_t st,
                        const float* resid = nullptr, const float* scale = nullptr, int act = 0,
                        const float* sa = nullptr, const float* sib = nullptr) {
    ConvArgs a; a.x = x; a.w = w; a.b = b; a.y = y; a.cin = cin; a.cout = cout; a.L = L; a.k = 1; a.dil = 1;
    a.resid = resid; a.scale = scale; a.act = act; a.snake_a = sa; a.snake_b = sib; a.wpk = packed_of(w); a.planes = tl_codec_planes;
    return launch_conv1d(a, st);
}
static hipError_t convk(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L, int k, int dil,
                        hipStream_t st, const float* sa = nullptr, const float* sib = nullptr, int act = 0) {
    ConvArgs a; a.x = x; a.w = w; a.b = b; a.y = y; a.cin = cin; a.cout = cout; a.L = L; a.k = k; a.dil = dil;
    a.snake_a = sa; a.snake_b = sib; a.act = act; a.wpk = packed_of(w); a.planes = tl_codec_planes;
    return launch_conv1d(a, st);
}

// frames already on device in ws.frames; result in ws.pcm. taps: host pointers or nullptr.
// c0 = 0: whole-utterance decode, ws.pcm = [T*spf]. c0 > 0 (segment decode): the front (quantiser, pre_conv,
// pre-transformer: everything with unbounded left context, and cheap) runs over all T frames, the convolutional
// stack only over frames [c0, T), and ws.pcm = [(T-c0)*spf]; samples of frames >= c0 + CODEC_CTX_FRAMES are
// identical to the whole-utterance decode (every kernel sums each output in a position-independent order).
q3_status codec_decode_dev(const q3_model* m, CodecWS& ws, int T, hipStream_t st, float** taps, int c0) {
    const q3_config& c = m->cfg;
    tl_codec_model = m;
    // Q3_CODEC_PLANES=2|3: A/B aid, overrides q3_model_set_codec_planes for the vocoder only (never the encoders)
    static const int env_planes = [] { const char* e = getenv("Q3_CODEC_PLANES"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 3) ? v : 0; }();
    const int NPL = env_planes ? env_planes : m->codec_planes;
    const CodecPlanesScope planes_scope(NPL);
    const int CD = c.dec_cb_dim, Q = c.dec_q_dim, LAT = c.dec_latent, DH = c.dec_hidden, QD = c.dec_heads * c.dec_head_dim, DI = c.dec_inter;
    auto TAP = [&](int id, const float* dev, size_t n) -> q3_status {
        if (taps && taps[id]) { HIPC(hipStreamSynchronize(st)); HIPC(q3_hipMemcpy(taps[id], dev, n * 4, hipMemcpyDeviceToHost)); }
        return Q3_OK;
    };
    float *A = ws.bufA, *B = ws.bufB, *C = ws.bufC, *D = ws.bufD, *E = ws.bufE;
    // D1 quantiser: E1 = A[0..256T), E2 = A[256T..512T) → quantized in B [Q][T]
    float* e1 = A; float* e2 = A + (size_t)CD * T;
    HIPC(launch_rvq_embed(ws.frames, T, m->first_cb, m->rest_cbs_dev, e1, e2, CD, c.dec_cb_size, st));
    HIPC(conv1(e1, m->first_proj, nullptr, B, CD, Q, T, st));
    HIPC(conv1(e2, m->rest_proj, nullptr, B, CD, Q, T, st, B));
    Q3C(TAP(Q3_DEC_QUANT, B, (size_t)Q * T));
    // D2 pre_conv → C [LAT][T]
    HIPC(convk(B, m->pre_w, m->pre_b, C, Q, LAT, T, 3, 1, st));
    Q3C(TAP(Q3_DEC_PRECONV, C, (size_t)LAT * T));
    // D3 pre-transformer. hidden Hd = D [DH][T]
    float* Hd = D;
    HIPC(conv1(C, m->inp_w, m->inp_b, Hd, LAT, DH, T, st));
    float* Nn = E;                                   // [DH][T]
    float* q = A; float* k = A + (size_t)QD * T; float* v = A + (size_t)2 * QD * T; float* ao = A + (size_t)3 * QD * T;
    float* g = B; float* u = B + (size_t)DI * T;
    const float scale = (float)pow((double)c.dec_head_dim, -0.5);
    for (int l = 0; l < c.dec_layers; ++l) {
        const DecLayerW& L = m->dl[l];
        HIPC(launch_rmsnorm_c(Hd, L.in_ln, Nn, DH, T, c.dec_eps, st));
        // q | k | v (and gate | up below) as ONE launch when their packed weights sit back to back in the arena (they are
        // packed in this order at finalize: concatenating A-operand tiles along the output channels is just adjacency) —
        // 3 x 160 workgroups of 29 us each become one grid of 480; per output the same arithmetic
        // Only the bf16x3 kernel reads the packed images; its fallbacks (Q3_CONV_F32 / Q3_CONV_VALU A/B switches, shapes
        // outside its divisibility rules) read the f32 tensor of ONE projection, so the fusion is tied to that path and to
        // the f32 tensors being adjacent as well (a fallback would otherwise run past the end of L.q / L.gate).
        auto adjacent = [&](const float* w0, const float* w1, int cout, int cin) {
            const char* p0 = (const char*)m->pk(w0); const char* p1 = (const char*)m->pk(w1);
            return p0 && p1 && p1 == p0 + packed_conv_w_bytes(cout, cin, 1) && w1 == w0 + (size_t)cout * cin &&
                   cout % 32 == 0 && cin % 16 == 0;
        };
        static const bool no_fuse_qkv = getenv("Q3_CODEC_NO_QKV_FUSE") != nullptr || getenv("Q3_CONV_F32") != nullptr ||
                                        getenv("Q3_CONV_VALU") != nullptr;      // A/B aids
        if (!no_fuse_qkv && adjacent(L.q, L.k, QD, DH) && adjacent(L.k, L.v, QD, DH)) {
            HIPC(conv1(Nn, L.q, nullptr, q, DH, 3 * QD, T, st));
        } else {
            HIPC(conv1(Nn, L.q, nullptr, q, DH, QD, T, st));
            HIPC(conv1(Nn, L.k, nullptr, k, DH, QD, T, st));
            HIPC(conv1(Nn, L.v, nullptr, v, DH, QD, T, st));
        }
        HIPC(launch_rope_c(q, k, ws.cs, ws.sn, c.dec_heads, c.dec_head_dim, T, st));
        HIPC(launch_attn_c(q, k, v, ao, c.dec_heads, c.dec_head_dim, T, scale, st));
        HIPC(conv1(ao, L.o, nullptr, Hd, QD, DH, T, st, Hd, L.attn_scale));
        HIPC(launch_rmsnorm_c(Hd, L.post_ln, Nn, DH, T, c.dec_eps, st));
        if (!no_fuse_qkv && adjacent(L.gate, L.up, DI, DH)) {
            HIPC(conv1(Nn, L.gate, nullptr, g, DH, 2 * DI, T, st));
        } else {
            HIPC(conv1(Nn, L.gate, nullptr, g, DH, DI, T, st));
            HIPC(conv1(Nn, L.up, nullptr, u, DH, DI, T, st));
        }
        HIPC(launch_silu_mul(g, u, g, (int64_t)DI * T, st));
        HIPC(conv1(g, L.down, nullptr, Hd, DI, DH, T, st, Hd, L.mlp_scale));
    }
    HIPC(launch_rmsnorm_c(Hd, m->dec_norm, Nn, DH, T, c.dec_eps, st));
    HIPC(conv1(Nn, m->outp_w, m->outp_b, C, DH, LAT, T, st));      // C [LAT][T]
    Q3C(TAP(Q3_DEC_PRETRANS, C, (size_t)LAT * T));
    // D4 upsample stages: cur in C (segment decode: the latent columns [c0, T) copied out to F)
    float* cur = C; float* o1 = A; float* o2 = B;
    int L = T;
    if (c0 > 0) {
        HIPC(launch_copy_rows(C + c0, T, ws.bufF, T - c0, LAT, T - c0, st));
        cur = ws.bufF; L = T - c0;
    }
    for (int i = 0; i < 2; ++i) {
        const UpW& U = m->up[i];
        float* upo = (cur == C) ? A : C;            // transconv output [LAT][L*r]
        HIPC(launch_transconv1d_taps(cur, U.tw, U.tb, upo, LAT, LAT, L, U.ratio, 1, nullptr, nullptr, st, nullptr, nullptr, nullptr, m->pk(U.tw), NPL));
        L *= U.ratio;
        // dwconv → LN → pw1+GELU → pw2·gamma + residual (in place into upo)
        float* dw = (upo == A) ? C : A;             // [LAT][L]
        HIPC(launch_dwconv7(upo, U.dww, U.dwb, dw, LAT, L, st));
        float* ln = dw + (size_t)LAT * L;           // second half of that buffer
        HIPC(launch_layernorm_c(dw, U.nw, U.nb, ln, LAT, L, 1e-6f, st));
        HIPC(conv1(ln, U.p1w, U.p1b, B, LAT, 4 * LAT, L, st, nullptr, nullptr, 1));
        HIPC(conv1(B, U.p2w, U.p2b, upo, 4 * LAT, LAT, L, st, upo, U.gamma));
        cur = upo;
        Q3C(TAP(Q3_DEC_UP0 + i, cur, (size_t)LAT * L));
    }
    (void)o1; (void)o2;
    // D5-D9. SnakeBeta is applied by the PRODUCER's epilogue (each element activated once, not once per
    // consuming output-channel tile): every tensor below exists as "raw" (residual / tap) and/or "act"
    // (= snake of the next consumer).
    int Cc = c.dec_dim;
    float* pool4[4] = {A, B, C, ws.bufF};
    auto other = [&](std::initializer_list<const float*> used) -> float* {
        for (float* p : pool4) { bool u = false; for (const float* q : used) if (q == p) u = true; if (!u) return p; }
        return nullptr;
    };
    // decoder.0 (k=7): raw only if tapped; activated with block 0's snake
    float* xact = other({cur});
    {
        ConvArgs a; a.x = cur; a.w = m->init_w; a.wpk = m->pk(m->init_w); a.b = m->init_b; a.cin = LAT; a.cout = Cc; a.L = L; a.k = 7; a.dil = 1; a.planes = NPL;
        a.post_a = m->blk[0].a; a.post_ib = m->blk[0].ib;
        if (taps && taps[Q3_DEC_INIT]) { float* raw = other({cur, xact}); a.y = raw; a.y2 = xact; HIPC(launch_conv1d(a, st)); Q3C(TAP(Q3_DEC_INIT, raw, (size_t)Cc * L)); }
        else { a.y = xact; HIPC(launch_conv1d(a, st)); }
    }
    static const int dils[3] = {1, 3, 9};
    for (int b = 0; b < 4; ++b) {
        const DecBlockW& Bk = m->blk[b];
        // transposed conv: raw Y (residual of unit 0) + YA = snake(act1 of unit 0)
        float* Y = other({xact});
        float* YA = other({xact, Y});
        HIPC(launch_transconv1d_taps(xact, Bk.tw, Bk.tb, Y, Bk.cin, Bk.cout, L, Bk.rate, 2, nullptr, nullptr, st, Bk.res[0].a1, Bk.res[0].ib1, YA, m->pk(Bk.tw), NPL));
        L *= Bk.rate; Cc = Bk.cout;
        float* T2 = other({Y, YA});
        for (int uu = 0; uu < 3; ++uu) {
            const ResUnitW& R = Bk.res[uu];
            const float* nxt_a = uu < 2 ? Bk.res[uu + 1].a1 : (b < 3 ? m->blk[b + 1].a : m->fin_a);
            const float* nxt_ib = uu < 2 ? Bk.res[uu + 1].ib1 : (b < 3 ? m->blk[b + 1].ib : m->fin_ib);
            {   // 96 / 192 channels: the whole unit in one launch (raw tensor updated in place, activated copy into T2)
                ResUnitArgs r{};
                r.xa = YA; r.y = Y; r.ya = T2; r.w1pk = m->pk(R.c1w); r.w2pk = m->pk(R.c2w); r.b1 = R.c1b; r.b2 = R.c2b;
                r.mid_a = R.a2; r.mid_ib = R.ib2; r.post_a = nxt_a; r.post_ib = nxt_ib; r.C = Cc; r.L = L; r.dil = dils[uu]; r.planes = NPL;
                const hipError_t e = launch_resunit(r, st);
                if (e == hipSuccess) { float* t = YA; YA = T2; T2 = t; continue; }
                if (e != hipErrorNotSupported) HIPC(e);
            }
            {   // conv7 (dilated) on the activated input; output activated with act2
                ConvArgs a; a.x = YA; a.w = R.c1w; a.wpk = m->pk(R.c1w); a.b = R.c1b; a.y = T2; a.cin = Cc; a.cout = Cc; a.L = L; a.k = 7; a.dil = dils[uu]; a.planes = NPL;
                a.post_a = R.a2; a.post_ib = R.ib2;
                HIPC(launch_conv1d(a, st));
            }
            {   // conv1 + residual: raw → Y (in place), activated → YA for the next consumer
                ConvArgs a; a.x = T2; a.w = R.c2w; a.wpk = m->pk(R.c2w); a.b = R.c2b; a.y = Y; a.y2 = YA; a.cin = Cc; a.cout = Cc; a.L = L; a.k = 1; a.dil = 1; a.resid = Y; a.planes = NPL;
                if (uu < 2) { a.post_a = Bk.res[uu + 1].a1; a.post_ib = Bk.res[uu + 1].ib1; }
                else if (b < 3) { a.post_a = m->blk[b + 1].a; a.post_ib = m->blk[b + 1].ib; }
                else { a.post_a = m->fin_a; a.post_ib = m->fin_ib; }
                HIPC(launch_conv1d(a, st));
            }
        }
        Q3C(TAP(Q3_DEC_BLK0 + b, Y, (size_t)Cc * L));
        xact = YA;
    }
    // D9 final conv on the activated tensor + clamp
    HIPC(convk(xact, m->fin_w, m->fin_b, ws.pcm, Cc, 1, L, 7, 1, st, nullptr, nullptr, 2));
    return Q3_OK;
}

extern "C" q3_status q3_decode_codes(q3_model* m, const uint32_t* frames_host, int n_frames, float* pcm_host, float** taps_host) {
    if (!m || !m->finalized) return set_err(Q3_INVALID_ARG, "model not finalized");
    if (n_frames < 0 || (n_frames > 0 && (!frames_host || !pcm_host))) return set_err(Q3_INVALID_ARG, "q3_decode_codes: bad argument");
    if (n_frames == 0) return Q3_OK;
    for (int f = 0; f < n_frames; ++f)
        for (int g = 1; g < 16; ++g)
            if (frames_host[(size_t)f * 16 + g] >= (uint32_t)m->cfg.dec_cb_size)
                return set_err(Q3_INVALID_ARG, "code %u out of range for codebook %d (frame %d)", frames_host[(size_t)f * 16 + g], g, f);
    HIPC(hipSetDevice(m->device));
    CodecWS ws;
    q3_status st = codec_reserve(m, ws, n_frames);
    if (st == Q3_OK) {
        hipError_t e = q3_hipMemcpy(ws.frames, frames_host, (size_t)n_frames * 16 * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) st = set_err(Q3_HIP_ERROR, "hipMemcpy frames: %s", hipGetErrorString(e));
    }
    if (st == Q3_OK) st = codec_decode_dev(m, ws, n_frames, 0, taps_host);
    if (st == Q3_OK) {
        hipError_t e = q3_hipDeviceSynchronize();
        if (e == hipSuccess) e = q3_hipMemcpy(pcm_host, ws.pcm, (size_t)n_frames * samples_per_frame(m->cfg) * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = set_err(Q3_HIP_ERROR, "decode: %s", hipGetErrorString(e));
    }
    ws.release();
    return st;
}

