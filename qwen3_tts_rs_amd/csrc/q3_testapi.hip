// q3_testapi.hip — low-level entry points of the parity tests, submission / profiling read-outs, bench.py roofline replays
// (one of the five units of the engine: q3_engine.h says which holds what)
#include "q3_engine.h"

extern "C" q3_status q3_talker_step(q3_session* s, const float* embeds_host, float* hidden_host, float* logits_host) {
    if (!s || !embeds_host) return set_err(Q3_INVALID_ARG, "null argument");
    if (!s->prefilled) return set_err(Q3_INVALID_ARG, "session not prefilled");
    const q3_config& c = s->m->cfg;
    HIPC(hipSetDevice(s->m->device));
    // the step writes K/V at `pos`: refuse BEFORE running when that slot does not exist (a step at pos == max_seq - 1 is valid)
    std::vector<int> posv(s->B);
    HIPC(sync_frames(s));
    HIPC(q3_hipMemcpy(posv.data(), s->pos, s->B * 4, hipMemcpyDeviceToHost));
    for (int p : posv) if (p >= s->max_seq) return set_err(Q3_KV_OVERFLOW, "KV cache full (%d)", s->max_seq);
    for (int b = 0; b < s->B; ++b) Q3C(kv_reserve_row(s, b, posv[(size_t)b] + 1));       // paged KV: the slot this step writes
    HIPC(hipMemcpyAsync(s->tb.X, embeds_host, (size_t)s->B * c.hidden * 4, hipMemcpyHostToDevice, s->stream));
    Q3C(talker_step(s, s->pos, 0, true));
    // advance positions by one (host-driven teacher forcing)
    HIPC(sync_frames(s));
    for (int& p : posv) p += 1;
    HIPC(q3_hipMemcpy(s->pos, posv.data(), s->B * 4, hipMemcpyHostToDevice));
    if (hidden_host) HIPC(q3_hipMemcpy(hidden_host, s->LASTH, (size_t)s->B * c.hidden * 4, hipMemcpyDeviceToHost));
    if (logits_host) HIPC(q3_hipMemcpy(logits_host, s->LOGITS, (size_t)s->B * c.codec_vocab * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_cp_generate(q3_session* s, const float* last_hidden_host, const float* sem_embed_host,
                                    uint32_t* codes15_host, float* cp_logits_host) {
    if (!s || !last_hidden_host || !codes15_host) return set_err(Q3_INVALID_ARG, "null argument");
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    HIPC(hipSetDevice(m->device));
    // the semantic embedding is looked up from tok on device; teacher forcing passes it as an embedding
    // row, so run pass 1 from an explicit buffer: temporarily stage it through CP_IN/cb.X.
    const int B = s->B, H = c.hidden, CH = c.cp_hidden, V = c.cp_vocab;
    HIPC(hipMemcpyAsync(s->LASTH, last_hidden_host, (size_t)B * H * 4, hipMemcpyHostToDevice, s->stream));
    const LmDims d = cp_dims(c);
    float* sem_dev = nullptr;
    if (sem_embed_host) { HIPC(hipMalloc((void**)&sem_dev, (size_t)B * H * 4)); HIPC(q3_hipMemcpy(sem_dev, sem_embed_host, (size_t)B * H * 4, hipMemcpyHostToDevice)); }
    q3_status st = Q3_OK;
    auto run = [&]() -> q3_status {
        for (int p = 0; p < c.n_groups; ++p) {
            CpGatherArgs g{};
            g.pass = p; g.last_hidden = s->LASTH; g.H = H; g.codec_emb = m->codec_emb; g.tok = s->tok;
            g.cp_emb = p >= 2 ? m->cp_emb[p - 2] : nullptr;
            g.cp_logits = p >= 2 ? s->CP_LOGITS + (size_t)(p - 2) * B * V : nullptr;
            g.cp_vocab = V; g.codes = s->codes; g.frame_idx = s->frame_idx; g.max_frames = s->max_frames; g.B = B;
            float* dst = m->mtp_w.t1 ? s->CP_IN : s->cb.X; const int ld = m->mtp_w.t1 ? H : CH;
            g.out = dst; g.ld_out = ld;
            if (p == 1 && sem_dev) HIPC(launch_copy_rows(sem_dev, H, dst, ld, B, H, s->stream));
            else HIPC(launch_cp_gather(g, s->stream));
            if (m->mtp_w.t1) {
                LinArgs a;
                a.N = CH; a.K = H; set_w(a, m->mtp_w, B, CH, H); a.x = s->CP_IN; a.ldx = H; a.bias = m->mtp_b; a.y = s->cb.X; a.ldy = CH; a.M = B; a.epi = EPI_NONE;
                HIPC(launch_linear(a, s->stream));
            }
            for (int i = 0; i < c.cp_layers; ++i)
                Q3C(lm_layer(s, d, m->cl[i], s->cb, s->ckcache + (size_t)i * s->ckv_layer_stride, s->cvcache + (size_t)i * s->ckv_layer_stride,
                             c.n_groups + 1, nullptr, p, 1));
            if (p >= 1) {
                LinArgs h;
                h.N = V; h.K = CH; set_w(h, m->cp_head[p - 1], B, V, CH); h.x = s->cb.X; h.ldx = CH; h.norm_w = m->cp_norm; h.eps = c.rms_eps;
                h.y = s->CP_LOGITS + (size_t)(p - 1) * B * V; h.ldy = V; h.M = B; h.epi = EPI_NONE;
                HIPC(launch_linear(h, s->stream));
            }
        }
        HIPC(sync_frames(s));
        return Q3_OK;
    };
    st = run();
    if (sem_dev) hipFree(sem_dev);
    Q3C(st);
    std::vector<float> lg((size_t)15 * B * V);
    HIPC(q3_hipMemcpy(lg.data(), s->CP_LOGITS, lg.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < 15; ++g) {
            const float* row = &lg[((size_t)g * B + b) * V];
            int best = 0; for (int i = 1; i < V; ++i) if (row[i] > row[best]) best = i;
            codes15_host[(size_t)b * 15 + g] = (uint32_t)best;
            if (cp_logits_host) memcpy(cp_logits_host + ((size_t)b * 15 + g) * V, row, (size_t)V * 4);
        }
    return Q3_OK;
}

extern "C" q3_status q3_frame_embed(q3_model* m, uint32_t sem_token, const uint32_t* codes15, const float* text_add_host, float* out_host) {
    if (!m || !m->finalized || !codes15 || !text_add_host || !out_host) return set_err(Q3_INVALID_ARG, "bad argument");
    const q3_config& c = m->cfg;
    HIPC(hipSetDevice(m->device));
    const int H = c.hidden, V = c.cp_vocab;
    if (sem_token >= (uint32_t)c.codec_vocab) return set_err(Q3_INVALID_ARG, "semantic token out of range");
    DevPool pool;
    float *rows, *logits, *out; uint32_t *tok, *codes; int *zero, *one;
    HIPC(pool.alloc(&rows, (size_t)H)); HIPC(pool.alloc(&logits, (size_t)V)); HIPC(pool.alloc(&out, (size_t)H));
    HIPC(pool.alloc(&tok, 1)); HIPC(pool.alloc(&codes, 16)); HIPC(pool.alloc(&zero, 1)); HIPC(pool.alloc(&one, 1));
    // codes 0..13 pre-written; code 14 enters through a one-hot logits row
    uint32_t frame[16] = {0};
    for (int g = 0; g < 14; ++g) { if (codes15[g] >= (uint32_t)V) return set_err(Q3_INVALID_ARG, "code out of range"); frame[1 + g] = codes15[g]; }
    if (codes15[14] >= (uint32_t)V) return set_err(Q3_INVALID_ARG, "code out of range");
    std::vector<float> lg((size_t)V, 0.0f); lg[codes15[14]] = 1.0f;
    const int h_one = 1;
    HIPC(q3_hipMemcpy(rows, text_add_host, (size_t)H * 4, hipMemcpyHostToDevice));
    HIPC(q3_hipMemcpy(logits, lg.data(), (size_t)V * 4, hipMemcpyHostToDevice));
    HIPC(q3_hipMemcpy(tok, &sem_token, 4, hipMemcpyHostToDevice));
    HIPC(q3_hipMemcpy(codes, frame, 64, hipMemcpyHostToDevice));
    HIPC(q3_hipMemcpy(one, &h_one, 4, hipMemcpyHostToDevice));
    FrameEmbedArgs f{};
    f.codec_emb = m->codec_emb; f.tok = tok; f.cp_logits_last = logits; f.cp_vocab = V;
    for (int g = 0; g < 15; ++g) f.cp_embs[g] = g < (int)m->cp_emb.size() ? m->cp_emb[(size_t)g] : nullptr;
    f.codes = codes; f.frame_idx = zero; f.max_frames = 1; f.text_rows = rows; f.trail_base = zero; f.trail_len = one; f.pad_row = zero;
    f.out = out; f.H = H; f.B = 1; f.n_acoustic = 15;
    HIPC(launch_frame_embed(f, 0));
    HIPC(q3_hipDeviceSynchronize());
    HIPC(q3_hipMemcpy(out_host, out, (size_t)H * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_sample(int device, const float* logits_host, const uint8_t* seen_host, const float* u_host, int rows, int vocab,
                               const q3_options* o, int token_count, uint32_t* tokens_host) {
    if (!logits_host || !u_host || !o || !tokens_host || rows < 1) return set_err(Q3_INVALID_ARG, "bad argument");
    if (vocab < 2 || vocab > 4096) return set_err(Q3_UNSUPPORTED, "vocab %d unsupported by the device sampler (2..4096)", vocab);
    HIPC(hipSetDevice(device));
    DevPool pool;
    float *lg, *u; uint8_t* seen = nullptr; uint32_t* tok;
    HIPC(pool.alloc(&lg, (size_t)rows * vocab)); HIPC(pool.alloc(&u, (size_t)rows)); HIPC(pool.alloc(&tok, (size_t)rows));
    HIPC(q3_hipMemcpy(lg, logits_host, (size_t)rows * vocab * 4, hipMemcpyHostToDevice));
    HIPC(q3_hipMemcpy(u, u_host, (size_t)rows * 4, hipMemcpyHostToDevice));
    if (seen_host) { HIPC(pool.alloc(&seen, (size_t)rows * vocab)); HIPC(q3_hipMemcpy(seen, seen_host, (size_t)rows * vocab, hipMemcpyHostToDevice)); }
    SampleArgs a; memset(&a, 0, sizeof a);
    a.logits = lg; a.ld = vocab; a.seen = seen; a.u = u; a.u_stride = 1; a.tok = tok; a.token_count_static = token_count < 0 ? 0 : token_count;
    a.vocab = vocab; a.B = rows;
    a.apply_temp = (o->temperature != 1.0 && o->temperature > 0.0) ? 1 : 0;
    a.inv_temp = (float)(1.0 / o->temperature);
    a.greedy = o->temperature < 0.01 ? 1 : 0;
    a.top_k = o->top_k; a.use_top_p = (o->top_p < 1.0 && o->top_p > 0.0) ? 1 : 0; a.top_p = (float)o->top_p;
    const bool pen = token_count >= 0;     // token_count < 0: plain `sample` without the penalty pipeline
    a.use_rep = (pen && seen && o->repetition_penalty != 1.0 && !(fabs(o->repetition_penalty - 1.0) < 1e-9)) ? 1 : 0;
    a.rep_pen = (float)o->repetition_penalty; a.rep_inv = 1.0f / (float)o->repetition_penalty;
    a.eos_id = pen ? o->eos_token_id : -1; a.min_new_tokens = pen ? o->min_new_tokens : 0; a.codec_eos = CODEC_EOS; a.use_suppress = pen ? 1 : 0;
    HIPC(launch_sample(a, 0));
    HIPC(q3_hipDeviceSynchronize());
    HIPC(q3_hipMemcpy(tokens_host, tok, (size_t)rows * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_fused_residual_rmsnorm(int device, int dtype, const void* x_host, const void* res_host, const void* w_host,
                                               int rows, int cols, float eps, void* normed_host, void* sum_host) {
    if (!x_host || !res_host || !w_host || !normed_host || !sum_host || rows < 1 || cols < 1) return set_err(Q3_INVALID_ARG, "bad argument");
    if (dtype != Q3_DTYPE_F32 && dtype != Q3_DTYPE_BF16) return set_err(Q3_UNSUPPORTED, "dtype %d unsupported", dtype);
    HIPC(hipSetDevice(device));
    const size_t es = dtype == Q3_DTYPE_F32 ? 4 : 2, n = (size_t)rows * cols;
    DevPool pool;
    char *x, *r, *w, *nm, *sm;
    HIPC(pool.alloc(&x, n * es)); HIPC(pool.alloc(&r, n * es)); HIPC(pool.alloc(&w, (size_t)cols * es)); HIPC(pool.alloc(&nm, n * es)); HIPC(pool.alloc(&sm, n * es));
    HIPC(q3_hipMemcpy(x, x_host, n * es, hipMemcpyHostToDevice)); HIPC(q3_hipMemcpy(r, res_host, n * es, hipMemcpyHostToDevice));
    HIPC(q3_hipMemcpy(w, w_host, (size_t)cols * es, hipMemcpyHostToDevice));
    if (dtype == Q3_DTYPE_F32) HIPC(launch_fused_residual_rmsnorm_f32((float*)x, (float*)r, (float*)w, (float*)nm, (float*)sm, rows, cols, eps, 0));
    else HIPC(launch_fused_residual_rmsnorm_bf16((uint16_t*)x, (uint16_t*)r, (uint16_t*)w, (uint16_t*)nm, (uint16_t*)sm, rows, cols, eps, 0));
    HIPC(q3_hipDeviceSynchronize());
    HIPC(q3_hipMemcpy(normed_host, nm, n * es, hipMemcpyDeviceToHost)); HIPC(q3_hipMemcpy(sum_host, sm, n * es, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_linear(int device, const float* x_host, const uint16_t* w_host, const float* bias_host, int M, int N, int K, float* y_host) {
    if (!x_host || !w_host || !y_host || M < 1 || N < 1 || K < 8 || K % 8) return set_err(Q3_INVALID_ARG, "bad argument (K must be a multiple of 8)");
    HIPC(hipSetDevice(device));
    DevPool pool;
    float *x, *y, *b = nullptr; uint16_t* w;
    // the engine's own choice (pick_mode): 4-row tiles for narrow projections at small M — not for short-K wide ones, not beyond 16 rows
    const int mode = (M <= 16 && N < 4096 && !short_k_wide(N, K) && (M <= 2 || (N <= 1024 && M <= 8))) ? 2 : 1;
    const size_t wt_elems = tiled_elems(mode, N, K);
    std::vector<uint16_t> wt(wt_elems);
    retile_bf16(w_host, N, K, wt.data(), mode);
    HIPC(pool.alloc(&x, (size_t)M * K)); HIPC(pool.alloc(&y, (size_t)M * N)); HIPC(pool.alloc(&w, wt_elems));
    HIPC(q3_hipMemcpy(x, x_host, (size_t)M * K * 4, hipMemcpyHostToDevice)); HIPC(q3_hipMemcpy(w, wt.data(), wt_elems * 2, hipMemcpyHostToDevice));
    if (bias_host) { HIPC(pool.alloc(&b, (size_t)N)); HIPC(q3_hipMemcpy(b, bias_host, (size_t)N * 4, hipMemcpyHostToDevice)); }
    const int step = mode == 1 ? Q3_MAX_BATCH : 16;          // up to 64 rows per launch on the 16-row tiles (wide-session kernels beyond 16)
    float* ws = nullptr; size_t ws_bytes = 0;
    if (M > 16 && N % 128 == 0 && K % 128 == 0) { ws_bytes = gemm_wide_ws_bytes(M < step ? M : step, N, K, EPI_NONE); HIPC(pool.alloc(&ws, ws_bytes / 4)); }
    for (int m0 = 0; m0 < M; m0 += step) {
        LinArgs a;
        a.W = w; a.N = N; a.K = K; a.x = x + (size_t)m0 * K; a.ldx = K; a.bias = b; a.y = y + (size_t)m0 * N; a.ldy = N; a.M = (M - m0) < step ? (M - m0) : step; a.epi = EPI_NONE;
        a.tiled = mode; a.Kpad = kpad_for(mode, K); a.ws = ws; a.ws_bytes = ws_bytes;
        HIPC(launch_linear(a, 0));
    }
    HIPC(q3_hipDeviceSynchronize());
    HIPC(q3_hipMemcpy(y_host, y, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

#ifdef Q3_TRACE
// development builds only (not declared in include/q3tts.h): arm the per-node stamp buffer BEFORE the first
// q3_session_generate (the captured graph keeps the slice pointers), then read the stamps of the last replayed frame.
extern "C" q3_status q3_debug_trace_enable(q3_session* s, int max_nodes) {
    if (!s || max_nodes < 1) return set_err(Q3_INVALID_ARG, "q3_debug_trace_enable");
    const size_t bytes = (size_t)max_nodes * TRACE_NODE * 8;
    HIPC(hipMalloc((void**)&s->trace_buf, bytes));
    HIPC(hipMemset(s->trace_buf, 0, bytes));
    s->trace_cap = max_nodes;
    return Q3_OK;
}
extern "C" q3_status q3_debug_trace_read(q3_session* s, unsigned long long* stamps_host, int* meta_host, int cap_nodes, int* n_nodes) {
    if (!s || !s->trace_buf || !n_nodes) return set_err(Q3_INVALID_ARG, "q3_debug_trace_read");
    HIPC(sync_frames(s));
    const int n = (int)s->trace_meta.size();
    *n_nodes = n;
    if (stamps_host && meta_host) {
        if (cap_nodes < n) return set_err(Q3_INVALID_ARG, "trace buffer too small");
        HIPC(q3_hipMemcpy(stamps_host, s->trace_buf, (size_t)n * TRACE_NODE * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const auto& t = s->trace_meta[i];
            const int v[7] = {t.kind, t.a, t.b, t.c, t.d, t.e, t.f};
            memcpy(meta_host + (size_t)i * 7, v, sizeof v);
        }
    }
    return Q3_OK;
}
#endif

extern "C" q3_status q3_session_submit_info(q3_session* s, int* path, int* nodes) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (path) *path = s->aql ? 1 + s->aql_mode : s->graph ? 1 : 0;
    if (nodes) *nodes = s->aql ? q3::aql_program_nodes(s->aql) : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_submit_fences(q3_session* s, int* acquire_free, int* release_free) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    q3::aql_program_fence_free(s->aql, acquire_free, release_free);
    return Q3_OK;
}

extern "C" q3_status q3_session_frame_bytes(q3_session* s, int kv_len, double* weight_bytes, double* kv_bytes) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    const q3_config& c = s->m->cfg;
    auto layer_params = [](int H, int I, int nh, int nkv) { return (double)H * (nh + 2 * nkv) * HEAD_DIM + (double)H * nh * HEAD_DIM + 3.0 * (double)H * I; };
    const double talker = c.n_layers * layer_params(c.hidden, c.inter, c.n_heads, c.n_kv_heads) + (double)c.codec_vocab * c.hidden;
    const double cp_pass = c.cp_layers * layer_params(c.cp_hidden, c.cp_inter, c.cp_heads, c.cp_kv_heads) + (double)c.cp_vocab * c.cp_hidden +
                           (c.hidden != c.cp_hidden ? (double)c.cp_hidden * c.hidden : 0.0);
    if (weight_bytes) *weight_bytes = 2.0 * (talker + 15.0 * cp_pass);          // bf16; SURVEY §8(d)
    if (kv_bytes) {
        const double kv_tok = 2.0 * c.n_kv_heads * HEAD_DIM * (s->kv_bf16 ? 2.0 : 4.0) * c.n_layers;   // f32 K/V (default) or a bf16 session's
        const double cp_tok = 2.0 * c.cp_kv_heads * HEAD_DIM * 4.0 * c.cp_layers;
        *kv_bytes = s->B * (kv_tok * kv_len + cp_tok * 135.0);
    }
    return Q3_OK;
}

// profiling read-out: accumulated GPU milliseconds / algorithmic bytes / launches of the bf16 GEMV family
extern "C" q3_status q3_session_profile_read(q3_session* s, double* ms, double* bytes, long* launches, int reset) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (ms) *ms = s->prof_linear.ms; if (bytes) *bytes = s->prof_linear.bytes; if (launches) *launches = s->prof_linear.launches;
    if (reset) s->prof_linear = ProfAcc();
    return Q3_OK;
}

// the distinct GEMV launches (and how often each ran) since profiling was enabled / last reset: rows of 8 ints
// {M, N, K, epilogue, fused input RMSNorm (0 / 1), reserved (0), tiling, count} — what q3_bench_linear can replay
extern "C" q3_status q3_session_profile_shapes(q3_session* s, int* rows, int cap_rows, int* n_rows, int reset) {
    if (!s || !n_rows) return set_err(Q3_INVALID_ARG, "null argument");
    *n_rows = (int)s->prof_shapes.size();
    if (rows) {
        if (cap_rows < *n_rows) return set_err(Q3_INVALID_ARG, "shape buffer too small (%d < %d rows)", cap_rows, *n_rows);
        for (int i = 0; i < *n_rows; ++i) {
            const ProfShape& q = s->prof_shapes[i];
            const int v[8] = {q.M, q.N, q.K, q.epi, q.rms, q.produce, q.tiled, q.count};
            memcpy(rows + (size_t)i * 8, v, sizeof v);
        }
    }
    if (reset) s->prof_shapes.clear();
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// micro-benchmark of one GEMV shape (kernel development aid, used by tools/bench_kernels.py):
// `iters` back-to-back launches cycling over `n_copies` distinct weight buffers (so the stream comes
// from HBM, not the 256 MiB Infinity Cache), captured in one hipGraph and timed with HIP events.
// epi: LinEpi; rms: fused input RMSNorm; tiled: 1 = 16-row MFMA tiles, 2 = 4-row tiles, 3 = 16-row tiles with split-K in two,
// 0 = first-generation VALU kernel, < 0 = the engine's choice for an unsplit launch.
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_bench_linear(int device, int M, int N, int K, int epi, int rms, int tiled, int iters, int n_copies,
                                     double* avg_us) {
    if (M < 1 || M > Q3_MAX_BATCH || N < 16 || K < 32 || iters < 1 || n_copies < 1 || !avg_us || epi < EPI_NONE || epi > EPI_SWIGLU || rms < 0 || rms > 1)
        return set_err(Q3_INVALID_ARG, "q3_bench_linear: bad argument (epi 0..3, rms 0/1)");
    if (tiled < 0) tiled = (M <= 16 && N < 4096 && !short_k_wide(N, K) && (M <= 2 || (N <= 1024 && M <= 8))) ? 2 : 1;    // the engine's choice (pick_mode)
    const bool sk2 = tiled == 3;          // 3 = 16-row tiles, split-K in two (LinArgs::ksplit): y alternates between two buffers,
    if (sk2) tiled = 1;                   // each launch clearing the other one as its side job, as the frame loop's neighbours do
    HIPC(hipSetDevice(device));
    DevPool pool;
    const size_t welems = tiled == 2 ? tiled_elems(2, N, K) : tiled_elems(1, N, K);
    const int nmat = epi == EPI_SWIGLU ? 2 : 1;
    uint16_t* w; float *x, *y, *nw, *res;
    HIPC(pool.alloc(&w, welems * nmat * n_copies));
    HIPC(pool.alloc(&x, (size_t)Q3_MAX_BATCH * K)); HIPC(pool.alloc(&y, (size_t)2 * Q3_MAX_BATCH * N)); HIPC(pool.alloc(&nw, (size_t)K)); HIPC(pool.alloc(&res, (size_t)Q3_MAX_BATCH * N));
    HIPC(hipMemset(y, 0, (size_t)2 * Q3_MAX_BATCH * N * 4));
    {   // random-ish bf16 weights / f32 activations (never zeros: DVFS, guide §5.4 rule 25)
        std::vector<uint16_t> hw(welems);
        q3_synth_fill(1, "bench.w", Q3_DTYPE_BF16, 0.02f, 0.0f, (int64_t)welems, hw.data());
        for (int c = 0; c < nmat * n_copies; ++c) HIPC(q3_hipMemcpy(w + (size_t)c * welems, hw.data(), welems * 2, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)Q3_MAX_BATCH * K), hn((size_t)K, 1.0f);
        q3_synth_fill(2, "bench.x", Q3_DTYPE_F32, 1.0f, 0.0f, (int64_t)hx.size(), hx.data());
        HIPC(q3_hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        HIPC(q3_hipMemcpy(nw, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st; HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* ws = nullptr; size_t ws_bytes = 0;
    if (M > 16 && tiled == 1 && N % 128 == 0 && K % 128 == 0) { ws_bytes = gemm_wide_ws_bytes(M, N, K, epi); HIPC(pool.alloc(&ws, ws_bytes / 4)); }
    auto one = [&](int i) -> hipError_t {
        LinArgs a;
        a.ws = ws; a.ws_bytes = ws_bytes;
        const int c = i % n_copies;
        a.W = w + (size_t)c * nmat * welems; a.W2 = nmat == 2 ? a.W + welems : nullptr;
        a.N = N; a.K = K; a.Kpad = tiled == 2 ? up128(K) : up32(K); a.tiled = tiled; a.x = x; a.ldx = K; a.y = y; a.ldy = N; a.M = M; a.epi = epi;
        if (rms) { a.norm_w = nw; a.eps = 1e-6f; }
        if (epi == EPI_RESID) { a.resid = res; a.ldr = N; }
        if (sk2) { a.ksplit = 2; a.y = y + (size_t)(i & 1) * Q3_MAX_BATCH * N; a.zero = y + (size_t)((i + 1) & 1) * Q3_MAX_BATCH * N; a.zero_n = M * N; }
        return launch_linear(a, st);
    };
    for (int i = 0; i < 4; ++i) HIPC(one(i));
    HIPC(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    HIPC(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    hipError_t e = hipSuccess;
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = one(i);
    hipError_t e2 = hipStreamEndCapture(st, &g);
    if (e != hipSuccess || e2 != hipSuccess) return set_err(Q3_HIP_ERROR, "bench capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIPC(hipGraphLaunch(ge, st)); HIPC(hipStreamSynchronize(st));     // warm
    hipEvent_t ev0, ev1; HIPC(hipEventCreate(&ev0)); HIPC(hipEventCreate(&ev1));
    double sum = 0.0;          // mean over 5 replays (not the best one)
    for (int rep = 0; rep < 5; ++rep) {
        HIPC(hipEventRecord(ev0, st));
        HIPC(hipGraphLaunch(ge, st));
        HIPC(hipEventRecord(ev1, st));
        HIPC(hipStreamSynchronize(st));
        float ms = 0; HIPC(hipEventElapsedTime(&ms, ev0, ev1));
        sum += ms;
    }
    *avg_us = sum / 5.0 * 1000.0 / iters;
    hipEventDestroy(ev0); hipEventDestroy(ev1); hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    return Q3_OK;
}


