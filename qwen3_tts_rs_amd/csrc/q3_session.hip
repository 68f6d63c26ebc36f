// q3_session.hip — sessions: KV paging, talker / code-predictor step, frame capture and submission, prefill, generate, streaming, run / decode / get
// (one of the five units of the engine: q3_engine.h says which holds what)
#include "q3_engine.h"

// Everything this session has in flight has landed: the frames on the library's own AQL queue (q3_aql.cpp; not ordered with any
// HIP stream) and whatever rides the session's stream. Every host-side wait of the engine goes through here.
hipError_t sync_frames(q3_session* s) {
    if (s->aql) {
        std::string why;
        if (!q3::aql_wait(s->aql, &why)) { s->aql_failed = true; set_err(Q3_HIP_ERROR, "AQL frame submission: %s", why.c_str()); return hipErrorUnknown; }
    }
    return hipStreamSynchronize(s->stream);
}

static hipError_t run_linear(q3_session* s, const LinArgs& a_in) {
    LinArgs a = a_in;
    a.ws = s->wide_ws; a.ws_bytes = s->wide_ws_bytes;
#ifdef Q3_TRACE
    a.trace = s->trace_next(0, a.M, a.N, a.K, a.epi, a.norm_w ? 1 : 0, a.tiled);
#endif
    if (!s->profile) return launch_linear(a, s->stream);
    // profiling: bracket the launch with event records (inside graph capture these become event-record
    // nodes, so the timestamps are taken on the GPU timeline without host launch latency in between)
    hipEvent_t e0, e1;
    if (s->prof_pool_next + 2 <= s->prof_pool.size()) { e0 = s->prof_pool[s->prof_pool_next++]; e1 = s->prof_pool[s->prof_pool_next++]; }
    else {
        hipError_t e = hipEventCreate(&e0); if (e != hipSuccess) return e;
        e = hipEventCreate(&e1); if (e != hipSuccess) return e;
        s->prof_pool.push_back(e0); s->prof_pool.push_back(e1); s->prof_pool_next = s->prof_pool.size();
    }
    hipError_t e = hipEventRecord(e0, s->stream); if (e != hipSuccess) return e;
    e = launch_linear(a, s->stream);
    hipError_t e2 = hipEventRecord(e1, s->stream);
    s->prof_events.push_back({e0, e1});
    s->prof_event_bytes.push_back((double)a.N * a.K * 2.0 * (a.epi == EPI_SWIGLU ? 2.0 : 1.0));
    {   // launch inventory (q3_session_profile_shapes): M, N, K, epilogue, fused input norm, reserved, tiling
        ProfShape ps{a.M, a.N, a.K, a.epi, a.norm_w ? 1 : 0, 0, a.ksplit == 2 ? 3 : a.tiled, 1};
        bool found = false;
        for (auto& q : s->prof_shapes)
            if (q.M == ps.M && q.N == ps.N && q.K == ps.K && q.epi == ps.epi && q.rms == ps.rms && q.produce == ps.produce && q.tiled == ps.tiled) { q.count += 1; found = true; break; }
        if (!found) s->prof_shapes.push_back(ps);
    }
    return e != hipSuccess ? e : e2;
}

// Paged KV bookkeeping (KvPool): row b gets the pages for positions [0, n_pos) it does not hold yet; the new table entries
// are queued on the session's stream ahead of the kernels that read them (hipMemcpyAsync stages pageable sources before it
// returns; kv_rows[b] never reallocates: reserved to KV_MAX_PAGES at creation).
q3_status kv_reserve_row(q3_session* s, int b, int n_pos) {
    if (!s->paged) return Q3_OK;
    if (n_pos > KV_MAX_PAGES * KV_PAGE_POS) return set_err(Q3_KV_OVERFLOW, "%d positions exceed a row's page table (%d)", n_pos, KV_MAX_PAGES * KV_PAGE_POS);
    std::vector<float*>& row = s->kv_rows[(size_t)b];
    const int need = (n_pos + KV_PAGE_POS - 1) / KV_PAGE_POS, have = (int)row.size();
    if (need <= have) return Q3_OK;
    KvPool& pool = s->kv_in_bf16 ? s->m->kv_pool16 : s->m->kv_pool;
    if (pool.take(need - have, row) != hipSuccess)
    {
        s->kv_overflow_row = b;
        return set_err(Q3_KV_OVERFLOW, "KV page pool exhausted: row %d needs %d more %s page(s) of %d positions (budget: %ld of %ld half-pages in use)",
                       b, need - have, s->kv_in_bf16 ? "bf16" : "f32", KV_PAGE_POS, s->m->kv_budget.used, s->m->kv_budget.limit);
    }
    static_assert(sizeof(float*) == sizeof(unsigned long long), "page table entries are 64-bit pointers");
    HIPC(hipMemcpyAsync(s->kv_table + (size_t)b * KV_MAX_PAGES + have, row.data() + have, (size_t)(need - have) * 8, hipMemcpyHostToDevice, s->stream));
    return Q3_OK;
}
// pages for what the next `frames` frames of every row can touch: frame f of a row writes position prefill_len + f, a row
// that reached its limit keeps rewriting position prefill_len + limit (k_sample freezes its counters)
q3_status kv_reserve_frames(q3_session* s, int frames) {
    if (!s->paged) return Q3_OK;
    s->kv_overflow_row = -1;
    for (int b = 0; b < s->B; ++b) {
        const SeqInfo& q = s->seq[(size_t)b];
        if (q.idle) continue;
        int upto = s->frames_run - q.start_run + frames;
        if (upto > q.limit) upto = q.limit;
        if (upto < 0) upto = 0;
        Q3C(kv_reserve_row(s, b, q.prefill_len + upto + 1));
    }
    return Q3_OK;
}
void kv_release_row(q3_session* s, int b) {        // the caller has drained every stream that may still touch the row
    if (s->paged && !s->kv_rows[(size_t)b].empty()) (s->kv_in_bf16 ? s->m->kv_pool16 : s->m->kv_pool).give(s->kv_rows[(size_t)b]);
}

// bf16 sessions: every row's f32 pages -> as many pages of the bf16 pool (k_kv_pages_to_bf16), the table rewritten, the f32
// pages returned. The caller has drained the stream; this drains it again before the f32 pages go back.
static q3_status kv_convert_to_bf16(q3_session* s) {
    if (!s->paged) return set_err(Q3_UNSUPPORTED, "bf16 K/V needs the paged cache");
    const q3_config& c = s->m->cfg;
    std::vector<unsigned long long> src, dst; std::vector<std::vector<float*>> fresh((size_t)s->B);
    q3_status st = Q3_OK;
    for (int b = 0; b < s->B && st == Q3_OK; ++b) {
        const int n = (int)s->kv_rows[(size_t)b].size();
        if (s->m->kv_pool16.take(n, fresh[(size_t)b]) != hipSuccess) { st = set_err(Q3_KV_OVERFLOW, "KV page pool (bf16) exhausted: row %d needs %d page(s)", b, n); break; }
        for (int i = 0; i < n; ++i) { src.push_back((unsigned long long)s->kv_rows[(size_t)b][(size_t)i]); dst.push_back((unsigned long long)fresh[(size_t)b][(size_t)i]); }
    }
    if (st != Q3_OK) { for (auto& f : fresh) if (!f.empty()) s->m->kv_pool16.give(f); return st; }
    const int n = (int)src.size();
    hipError_t e = hipMemcpyAsync(s->kv_conv, src.data(), (size_t)n * 8, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->kv_conv + (size_t)s->B * KV_MAX_PAGES, dst.data(), (size_t)n * 8, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = launch_kv_pages_to_bf16(s->kv_conv, s->kv_conv + (size_t)s->B * KV_MAX_PAGES, n, c.n_layers, c.n_kv_heads, s->m->kv_pool.layer_stride(), s->m->kv_pool.v_delta(), s->stream);
    if (e == hipSuccess) e = sync_frames(s);
    if (e != hipSuccess) { for (auto& f : fresh) if (!f.empty()) s->m->kv_pool16.give(f); return set_err(Q3_HIP_ERROR, "K/V conversion to bf16: %s", hipGetErrorString(e)); }
    for (int b = 0; b < s->B; ++b) {
        s->m->kv_pool.give(s->kv_rows[(size_t)b]);
        s->kv_rows[(size_t)b].assign(fresh[(size_t)b].begin(), fresh[(size_t)b].end());
        if (!s->kv_rows[(size_t)b].empty())
            HIPC(hipMemcpyAsync(s->kv_table + (size_t)b * KV_MAX_PAGES, s->kv_rows[(size_t)b].data(), s->kv_rows[(size_t)b].size() * 8, hipMemcpyHostToDevice, s->stream));
    }
    HIPC(sync_frames(s));
    s->kv_in_bf16 = true;
    return Q3_OK;
}

// one DecoderLayer (transformer.rs:442-467) for the single new token of every sequence
q3_status lm_layer(q3_session* s, const LmDims& d, const LayerW& w, LmBuf& b, float* kc, float* vc, int max_seq,
                   const int* pos_dev, int pos_static, int n_splits, int rows_per_seq, bool skip_qkv,
                   const CpGatherArgs* fold,         // fold: this layer's attention does the pass's gather
                   int paged_layer) {                // >= 0: the talker's paged cache, this layer's index (kc / vc / max_seq unused)
    const q3_model* m = s->m;
    // B = number of activation ROWS of this step: one per sequence, or rows_per_seq consecutive positions per
    // sequence (chunked prefill, the code predictor's 2-token first pass)
    const int QD = d.nh * HEAD_DIM, KD = d.nkv * HEAD_DIM, B = s->B * rows_per_seq;
    LinArgs a;
    a.N = QD + 2 * KD; a.K = d.H; set_w(a, w.qkv, B, a.N, a.K); a.x = b.X; a.ldx = d.H; a.norm_w = w.in_ln; a.eps = d.eps;
    a.y = b.QKV; a.ldy = QD + 2 * KD; a.M = B; a.epi = EPI_NONE;
    // Wide sessions: the q|k|v GEMM hands its K-slice sums straight to the attention kernel, which adds them on the 3 x 128
    // values it needs (AttnArgs::qkv_part) — the slice-sum launch in between (93 per frame at 1.7B, 5.2 us + gap each) is gone.
    // Only for the one-launch decode attention (static one-row steps); Q3_WIDE_NO_QKV_FUSE=1: off (A/B aid)
    static const bool qkv_fuse = getenv("Q3_WIDE_NO_QKV_FUSE") == nullptr;
    WidePartial wp{nullptr, nullptr, 0};
    if (!skip_qkv) {       // skip: the caller already filled b.QKV (code predictor layer 0, table rows)
        bool fused = false;
        if (qkv_fuse && B >= gemm_wide_min_rows() && rows_per_seq == 1 && !s->legacy_attn && !s->profile && !s->debug && s->wide_ws) {
            LinArgs a2 = a; a2.ws = s->wide_ws; a2.ws_bytes = s->wide_ws_bytes;
            const hipError_t e = launch_gemm_wide_partial(a2, s->stream, &wp);
            if (e == hipSuccess && wp.S <= 8) fused = true;
            else if (e != hipSuccess && e != hipErrorNotSupported) HIPC(e);
            else if (e == hipSuccess) { wp = WidePartial{nullptr, nullptr, 0}; }       // more than eight slices: let the slice-sum launch redo it
        }
        if (!fused) { wp = WidePartial{nullptr, nullptr, 0}; HIPC(run_linear(s, a)); }
    }
    AttnArgs t{};
    t.qkv = b.QKV; t.ld_qkv = QD + 2 * KD; t.q_norm_w = w.q_norm; t.k_norm_w = w.k_norm; t.eps = d.eps;
    t.rope_cos = m->rope_cos; t.rope_sin = m->rope_sin; t.pos_dev = pos_dev; t.pos_static = pos_static;
    t.kcache = kc; t.vcache = vc; t.max_seq = max_seq; t.qbuf = b.Q; t.part = b.PART; t.out = b.ATT; t.ld_out = QD;
    t.B = B; t.nh = d.nh; t.nkv = d.nkv; t.n_splits = n_splits; t.rows_per_seq = rows_per_seq;
    if (paged_layer >= 0) {
        t.kv_pages = s->kv_table; t.kv_layer_off = (size_t)paged_layer * s->m->kv_pool.layer_stride();
        t.kv_vdelta = s->m->kv_pool.v_delta();
        t.kv_row_pages = (max_seq + KV_PAGE_POS - 1) / KV_PAGE_POS;
        t.kv_bf16 = s->kv_in_bf16 ? 1 : 0;
    }
    if (wp.part) { t.qkv_part = wp.part; t.qkv_ssq = wp.ssq; t.qkv_S = wp.S; t.qkv_K = d.H; t.qkv_eps = d.eps; }
    // Split-K projections (LinArgs::ksplit, k_gemv_sk2): o-proj and down-proj with N <= 2048 and K >= 2048 at 3 .. 16 rows (wide
    // sessions: blocks of 16 rows, see below)
    // run as two K halves that meet in the output through order-independent atomic adds. The output buffer must hold zeros:
    // SUM is cleared by this layer's attention launch (its last reader was the previous down-proj), X by the gate/up
    // launch (its last reader is this layer's o-proj, as the residual).
    const bool first2 = !s->legacy_attn && rows_per_seq == 2 && !pos_dev && pos_static == 0 && paged_layer < 0;
    const bool attn3 = !first2 && (s->legacy_attn || rows_per_seq > 1);
    // Wide sessions (B > 16, round 3): the same kernel over blocks of 16 rows (grid plane z) — one launch instead of the
    // split-K GEMM + slice-sum pair for exactly the narrow outputs where the second launch hurt most (B = 64, code
    // predictor o / down: 9.4 + 4.9 us -> one launch). Q3_WIDE_NO_SK2=1: off (A/B aid)
    static const bool wide_sk2 = getenv("Q3_WIDE_NO_SK2") == nullptr;
    const bool sk_rows = s->ksplit && B >= 3 && (B <= 16 || (wide_sk2 && B <= Q3_MAX_BATCH && rows_per_seq == 1)) && d.H <= 2048 && d.H % 4 == 0;
    // (round 6: also behind the code predictor's 2-token first pass — 16 rows at B = 8 — whose o-projection ran on k_gemv_lds with
    // 64 workgroups: 5.8 us against 3.6 as two K halves; Q3_FIRST2_NO_SK=1: the old kernel, A/B aid)
    static const bool first2_sk = getenv("Q3_FIRST2_NO_SK") == nullptr;
    const bool o_sk = sk_rows && (!first2 || first2_sk) && !attn3 && w.o.t1 && QD >= 2048 && up32(QD) / 32 >= 16;
    // (beyond 32 rows the 25 MB talker down-proj is re-read by every 16-row block — 32.0 us at B = 64 against 16.5 + 4.8 for
    // the GEMM pair — while the smaller matrices win: code predictor o 14.6 -> 6.8, down 15.7 -> 9.2, talker o 14.4 -> 11.7 us)
    const bool dn_big = B > 32 && (size_t)d.I * d.H * 2 > ((size_t)12 << 20);
    const bool dn_sk = sk_rows && !dn_big && w.down.t1 && w.gate.t1 && d.I >= 2048 && up32(d.I) / 32 >= 16;
    if (t.kv_bf16 && (first2 || attn3)) return set_err(Q3_UNSUPPORTED, "multi-row talker steps are not available once a session's K/V is bf16");
    if (first2) {
        if (fold) {                                 // pass-1 gather folded: row 2b+1 = table row tok[b]
            t.g_tok = fold->tok; t.g_qkv_tab = fold->qkv_tab; t.g_proj_tab = fold->proj_tab; t.g_proj_dim = fold->proj_dim;
            t.g_x = fold->out; t.g_ldx = fold->ld_out;
        }
        if (o_sk) { t.zero = b.SUM; t.zero_n = B * d.H; }
        HIPC(launch_attn_first2(t, s->stream));    // the code predictor's 2-token first pass
    } else if (attn3) {     // rows of one sequence depend on each other's K/V: three launches
        HIPC(launch_qknorm_rope_kv(t, s->stream));
        HIPC(launch_attn_decode(t, s->stream));
        HIPC(launch_attn_merge(t, s->stream));
    } else {
        if (fold) {
            t.g_logits = fold->cp_logits; t.g_vocab = fold->cp_vocab; t.g_qkv_tab = fold->qkv_tab;
            t.g_proj_tab = fold->proj_tab; t.g_proj_dim = fold->proj_dim; t.g_x = fold->out; t.g_ldx = fold->ld_out;
            t.g_codes = fold->codes; t.g_frame_idx = fold->frame_idx; t.g_max_frames = fold->max_frames; t.g_code_slot = fold->pass - 1;
        }
        if (o_sk) { t.zero = b.SUM; t.zero_n = B * d.H; }
        if (s->cp_attn && attn_cp_ok(t)) {      // <= 16 positions, static position: the code predictor
#ifdef Q3_TRACE
            t.trace = s->trace_next(1, t.B, t.nh, 1, t.pos_static, t.g_logits ? 1 : 0);
#endif
            HIPC(launch_attn_cp(t, s->stream));
        } else {
#ifdef Q3_TRACE
            t.trace = s->trace_next(2, t.B, t.nh, n_splits, t.pos_static, t.g_logits ? 1 : 0);
#endif
            HIPC(launch_attn_fused(t, s->stream));
#ifdef Q3_TRACE
            if (n_splits > 1) t.trace = s->trace_next(3, t.B, t.nh, n_splits, t.pos_static);
#endif
            if (n_splits > 1) HIPC(launch_attn_merge(t, s->stream));
        }
    }
    auto force16 = [&](LinArgs& l, const TW& tw) { l.tiled = 1; l.W = tw.t1; l.Kpad = kpad_for(1, l.K); l.ksplit = 2; };
    LinArgs o;
    o.N = d.H; o.K = QD; set_w(o, w.o, B, o.N, o.K); o.x = b.ATT; o.ldx = QD; o.resid = b.X; o.ldr = d.H; o.y = b.SUM; o.ldy = d.H; o.M = B; o.epi = EPI_RESID;
    if (o_sk) force16(o, w.o);
    HIPC(run_linear(s, o));
    LinArgs g;
    g.N = d.I; g.K = d.H; set_w2(g, w.gate, w.up, B, g.N, g.K); g.x = b.SUM; g.ldx = d.H; g.norm_w = w.post_ln; g.eps = d.eps;
    g.y = b.ACT; g.ldy = d.I; g.M = B; g.epi = EPI_SWIGLU;
    if (dn_sk) { g.zero = b.X; g.zero_n = B * d.H; }
    HIPC(run_linear(s, g));
    LinArgs dn;
    dn.N = d.H; dn.K = d.I; set_w(dn, w.down, B, dn.N, dn.K); dn.x = b.ACT; dn.ldx = d.I; dn.resid = b.SUM; dn.ldr = d.H; dn.y = b.X; dn.ldy = d.H; dn.M = B; dn.epi = EPI_RESID;
    if (dn_sk) force16(dn, w.down);
    HIPC(run_linear(s, dn));
    return Q3_OK;
}

static bool d_nh_ok(const q3_config& c) {     // GEMM prefill needs N % 64 == 0 for every projection and a 1- or 2-way GQA ratio
    const int qkv = (c.n_heads + 2 * c.n_kv_heads) * HEAD_DIM, rep = c.n_heads / (c.n_kv_heads ? c.n_kv_heads : 1);
    return qkv % 64 == 0 && c.hidden % 64 == 0 && c.inter % 64 == 0 && (rep == 1 || rep == 2);
}
LmDims talker_dims(const q3_config& c) { return LmDims{c.hidden, c.inter, c.n_heads, c.n_kv_heads, c.n_layers, c.rms_eps}; }
LmDims cp_dims(const q3_config& c) { return LmDims{c.cp_hidden, c.cp_inter, c.cp_heads, c.cp_kv_heads, c.cp_layers, c.rms_eps}; }

// talker layers on the contents of tb.X at position pos (device array or static); with_head: final
// norm → LASTH and codec_head → LOGITS (talker.rs:716-736)
q3_status talker_step(q3_session* s, const int* pos_dev, int pos_static, bool with_head, int rows_per_seq) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const LmDims d = talker_dims(c);
    for (int i = 0; i < c.n_layers; ++i)
        Q3C(lm_layer(s, d, m->tl[i], s->tb, s->paged ? nullptr : s->kcache + (size_t)i * s->kv_layer_stride,
                     s->paged ? nullptr : s->vcache + (size_t)i * s->kv_layer_stride,
                     s->max_seq, pos_dev, pos_static, s->n_splits, rows_per_seq, false, nullptr, s->paged ? i : -1));
    if (with_head) {
        // final norm of each sequence's LAST row of the step
        HIPC(launch_rmsnorm(s->tb.X + (size_t)(rows_per_seq - 1) * c.hidden, rows_per_seq * c.hidden, m->norm, s->LASTH, c.hidden, s->B,
                            c.hidden, c.rms_eps, s->stream));
        LinArgs h;
        h.N = c.codec_vocab; h.K = c.hidden; set_w(h, m->codec_head, s->B, h.N, h.K); h.x = s->LASTH; h.ldx = c.hidden; h.y = s->LOGITS; h.ldy = c.codec_vocab;
        h.M = s->B; h.epi = EPI_NONE;
        HIPC(run_linear(s, h));
    }
    return Q3_OK;
}

// generate_acoustic_codes (code_predictor.rs:320-416) as 16 single-token passes: pass 0 = talker
// hidden (pos 0), pass 1 = semantic embedding (pos 1) → lm_head[0]; pass p = embedding of code p-2
// (pos p) → lm_head[p-1]. (The reference runs passes 0 and 1 as one 2-token causal prefill; for
// causal attention that is the same computation.) Codes 0..13 are recorded by the next pass's
// gather, code 14 by frame_embed / the caller.
static q3_status cp_run(q3_session* s) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const LmDims d = cp_dims(c);
    const int H = c.hidden, CH = c.cp_hidden, V = c.cp_vocab, B = s->B;
    const int n_pass = c.n_groups;   // 16
    // First pass as the reference does it (code_predictor.rs:337-367): the talker hidden state and the semantic
    // embedding go through the layers TOGETHER as a 2-token causal prefill (rows 2b, 2b+1), when 2B rows fit.
    const bool two = !s->no_chunk && 2 * B <= 16;
    for (int p = two ? 1 : 0; p < n_pass; ++p) {
        const int rows = (two && p == 1) ? 2 : 1;
        CpGatherArgs g{};
        g.pass = p; g.last_hidden = s->LASTH; g.H = H; g.codec_emb = m->codec_emb; g.tok = s->tok;
        g.cp_emb = p >= 2 ? m->cp_emb[p - 2] : nullptr;
        g.cp_logits = p >= 2 ? s->CP_LOGITS + (size_t)(p - 2) * B * V : nullptr;
        g.cp_vocab = V; g.codes = s->codes; g.frame_idx = s->frame_idx; g.max_frames = s->max_frames; g.B = B;
        float* dst = m->mtp_w.t1 ? s->CP_IN : s->cb.X; const int ld = m->mtp_w.t1 ? H : CH;
        // 1.7B with pre-projected tables: only the talker hidden state (pass 0) still goes through the 2048 -> 1024
        // projection at run time; every embedding row arrives already projected (14 GEMV launches less per frame)
        const bool tabs = m->mtp_w.t1 && m->proj_tabs && s->proj_tables;
        auto project = [&](int M, int ldy, const float* x_in = nullptr) -> q3_status {
            LinArgs a;
            a.N = CH; a.K = H; set_w(a, m->mtp_w, M, CH, H); a.x = x_in ? x_in : s->CP_IN; a.ldx = H; a.bias = m->mtp_b; a.y = s->cb.X; a.ldy = ldy;
            a.M = M; a.epi = EPI_NONE;
            HIPC(run_linear(s, a));
            return Q3_OK;
        };
        // layer-0 q|k|v of a table row comes from the table too (both model sizes): the layer-0 qkv GEMV then only runs
        // for the rows that carry the talker hidden state
        const bool qt = m->qkv0_tabs && s->qkv_tables && (tabs || !m->mtp_w.t1);
        const int QKVD = (d.nh + 2 * d.nkv) * HEAD_DIM;
        bool skip0 = false, fold0 = false;
        auto qkv_rows0 = [&](int ldx, int ldy) -> q3_status {      // run-time layer-0 qkv of the B pass-0 rows
            LinArgs a;
            a.N = QKVD; a.K = CH; set_w(a, m->cl[0].qkv, B, QKVD, CH); a.x = s->cb.X; a.ldx = ldx; a.norm_w = m->cl[0].in_ln; a.eps = d.eps;
            a.y = s->cb.QKV; a.ldy = ldy; a.M = B; a.epi = EPI_NONE;
            HIPC(run_linear(s, a));
            return Q3_OK;
        };
        if (tabs) {
            if (rows == 2) {
                // B rows of talker hidden: projected straight from LASTH (the copy to CP_IN was a launch of its own)
                static const bool no_fold2 = getenv("Q3_CP_NO_FOLD") != nullptr;
                if (no_fold2) { g.pass = 0; g.out = s->CP_IN; g.ld_out = H; HIPC(launch_cp_gather(g, s->stream)); }
                Q3C(project(B, 2 * CH, no_fold2 ? s->CP_IN : s->LASTH));           // -> cb.X rows 2b
                g.pass = 1; g.out = s->cb.X + CH; g.ld_out = 2 * CH; g.proj_tab = m->sem_proj; g.proj_dim = CH;   // rows 2b+1
                if (qt) { g.qkv_tab = m->sem_qkv0; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV + QKVD; g.ld_qkv_out = 2 * QKVD; }
                // the semantic row (table row tok[b]) is read by the 2-token attention itself
                fold0 = qt && !s->legacy_attn && !no_fold2 && CH % 4 == 0;
                if (!fold0) HIPC(launch_cp_gather(g, s->stream));
                if (qt) { Q3C(qkv_rows0(2 * CH, 2 * QKVD)); skip0 = true; }
            } else if (p == 0) {
                g.out = s->CP_IN; g.ld_out = H;
                HIPC(launch_cp_gather(g, s->stream));
                Q3C(project(B, CH));
            } else {
                g.out = s->cb.X; g.ld_out = CH; g.proj_tab = p == 1 ? m->sem_proj : m->cp_proj[p - 2]; g.proj_dim = CH;
                if (qt) { g.qkv_tab = p == 1 ? m->sem_qkv0 : m->cp_qkv0[p - 2]; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV; g.ld_qkv_out = QKVD; skip0 = true; }
                // passes >= 2 with both tables: no launch of its own — the layer-0 attention re-derives the argmax and
                // reads the table rows itself (14 launches less per frame; Q3_CP_NO_FOLD=1: the separate launch, A/B aid)
                static const bool no_fold = getenv("Q3_CP_NO_FOLD") != nullptr;
                fold0 = qt && p >= 2 && !s->legacy_attn && !no_fold && CH % 4 == 0;
                if (!fold0) HIPC(launch_cp_gather(g, s->stream));
            }
        } else {
        if (rows == 2) {
            // row 2b = talker hidden (pass-0 source), row 2b+1 = semantic embedding (pass-1 source)
            g.pass = 0; g.out = dst; g.ld_out = 2 * ld;
            HIPC(launch_cp_gather(g, s->stream));
            g.pass = 1; g.out = dst + ld;
            if (qt) { g.qkv_tab = m->sem_qkv0; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV + QKVD; g.ld_qkv_out = 2 * QKVD; }
            HIPC(launch_cp_gather(g, s->stream));
            if (qt) { Q3C(qkv_rows0(2 * CH, 2 * QKVD)); skip0 = true; }
        } else {
            g.out = dst; g.ld_out = ld;
            if (qt && p >= 1) { g.qkv_tab = p == 1 ? m->sem_qkv0 : m->cp_qkv0[p - 2]; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV; g.ld_qkv_out = QKVD; skip0 = true; }
            HIPC(launch_cp_gather(g, s->stream));
        }
        if (m->mtp_w.t1) Q3C(project(B * rows, CH));
        }
        for (int i = 0; i < c.cp_layers; ++i)
            Q3C(lm_layer(s, d, m->cl[i], s->cb, s->ckcache + (size_t)i * s->ckv_layer_stride, s->cvcache + (size_t)i * s->ckv_layer_stride,
                         n_pass + 1, nullptr, rows == 2 ? 0 : p, 1, rows, i == 0 && skip0, (i == 0 && fold0) ? &g : nullptr));
        if (p >= 1) {
            LinArgs h;
            h.N = V; h.K = CH; set_w(h, m->cp_head[p - 1], B, V, CH); h.x = s->cb.X + (size_t)(rows - 1) * CH; h.ldx = rows * CH;
            h.norm_w = m->cp_norm; h.eps = c.rms_eps;
            h.y = s->CP_LOGITS + (size_t)(p - 1) * B * V; h.ldy = V; h.M = B; h.epi = EPI_NONE;
            HIPC(run_linear(s, h));
        }
    }
    return Q3_OK;
}

// sampling options of one request -> the sampler's per-row record (sampling.rs:140-319 branch conditions)
SampleRow sample_row(const q3_options& o) {
    SampleRow r{};
    r.apply_temp = (o.temperature != 1.0 && o.temperature > 0.0) ? 1 : 0;
    r.inv_temp = (float)(1.0 / o.temperature);
    r.greedy = o.temperature < 0.01 ? 1 : 0;
    r.top_k = o.top_k; r.use_top_p = (o.top_p < 1.0 && o.top_p > 0.0) ? 1 : 0; r.top_p = (float)o.top_p;
    r.use_rep = (o.repetition_penalty != 1.0 && !(fabs(o.repetition_penalty - 1.0) < 1e-9)) ? 1 : 0;
    r.rep_pen = (float)o.repetition_penalty; r.rep_inv = 1.0f / (float)o.repetition_penalty;
    r.eos_id = o.eos_token_id; r.min_new_tokens = o.min_new_tokens;
    return r;
}

static void fill_sample_args(q3_session* s, SampleArgs& a) {
    const q3_config& c = s->m->cfg;
    memset(&a, 0, sizeof a);
    a.logits = s->LOGITS; a.ld = c.codec_vocab; a.seen = s->seen; a.u = s->U; a.u_stride = s->max_frames + 2; a.limit = s->limit;
    a.draw_idx = s->token_count; a.tok = s->tok; a.token_count = s->token_count; a.frame_idx = s->frame_idx; a.pos = s->pos;
    a.vocab = c.codec_vocab; a.B = s->B;
    const SampleRow r = sample_row(s->opts);          // scalar fields = the first request's (every row reads its own through a.rows)
    a.apply_temp = r.apply_temp; a.inv_temp = r.inv_temp; a.greedy = r.greedy; a.top_k = r.top_k; a.use_top_p = r.use_top_p; a.top_p = r.top_p;
    a.use_rep = r.use_rep; a.rep_pen = r.rep_pen; a.rep_inv = r.rep_inv; a.eos_id = r.eos_id; a.min_new_tokens = r.min_new_tokens;
    a.rows = s->sample_rows;
    a.codec_eos = CODEC_EOS; a.use_suppress = 1;
    if (s->debug && s->logits_hist) { a.logits_hist = s->logits_hist; a.hist_stride_b = (s->max_frames + 1) * c.codec_vocab; a.hist_cap = s->max_frames + 1; }
}

// one frame of generate_codes (lib.rs:580-652)
static q3_status frame_launch(q3_session* s) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
#ifdef Q3_TRACE
    s->trace_node = 0; s->trace_meta.clear();        // every frame (and the capture) re-uses the same slices
#endif
    Q3C(cp_run(s));
    FrameEmbedArgs f{};
    f.codec_emb = m->codec_emb; f.tok = s->tok;
    for (int g = 0; g < 15; ++g) f.cp_embs[g] = g < (int)m->cp_emb.size() ? m->cp_emb[(size_t)g] : nullptr;
    f.cp_logits_last = s->CP_LOGITS + (size_t)14 * s->B * c.cp_vocab; f.cp_vocab = c.cp_vocab;
    f.codes = s->codes; f.frame_idx = s->frame_idx; f.max_frames = s->max_frames;
    f.text_rows = s->rows; f.trail_base = s->trail_base; f.trail_len = s->trail_len; f.pad_row = s->pad_row;
    f.out = s->tb.X; f.H = c.hidden; f.B = s->B; f.n_acoustic = c.n_groups - 1;
    HIPC(launch_frame_embed(f, s->stream));
    if (s->debug && s->cp_logits_hist) {
        // capture is host-indexed: only valid outside graph replay (debug sessions never use graphs)
        HIPC(hipMemcpyAsync(s->cp_logits_hist + (size_t)s->frames_run * 15 * s->B * c.cp_vocab, s->CP_LOGITS,
                            (size_t)15 * s->B * c.cp_vocab * 4, hipMemcpyDeviceToDevice, s->stream));
    }
    Q3C(talker_step(s, s->pos, 0, true));
    SampleArgs a; fill_sample_args(s, a); a.advance = 1;
    HIPC(launch_sample(a, s->stream));
    return Q3_OK;
}


// prefill positions and frame limit of a request as session_create will resolve them (talker.rs:451-491 / 511-564 / 585-627; the
// ICL length cap of lib.rs:913-929) — for callers that must know a row's worst-case KV extent before a session exists
void request_shape(const q3_request& r, int* prefill_len, int* limit) {
    const bool icl = r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes && r.ref_text_ids;
    const int n_ins = r.mode == Q3_MODE_VOICE_DESIGN ? r.n_instruct : 0;
    const int overlay = r.mode == Q3_MODE_VOICE_DESIGN ? 5 : 6;
    *prefill_len = n_ins + 3 + overlay + ((r.n_text > 0 && !icl) ? 1 : 0) + (icl ? r.n_ref + 1 : 0);
    int lim = r.opts.max_length;
    if (icl) { int cap = 6 * r.n_text; if (cap < 75) cap = 75; if (lim > cap) lim = cap; }
    *limit = lim;
}
// worst-case cost of a row against the model's KV budget (KvBudget units): every page prompt + limit + 1 positions can reach; a
// bf16 row holds f32 pages for its prompt and, while they are converted, bf16 pages beside them
long row_worst_units(int prefill_len, int limit, bool bf16) {
    const long full = (prefill_len + limit + 1 + KV_PAGE_POS - 1) / KV_PAGE_POS, pre = (prefill_len + 1 + KV_PAGE_POS - 1) / KV_PAGE_POS;
    return bf16 ? std::max(3 * pre, full) : 2 * full;
}
// a request that only occupies a row: a one-token CustomVoice prompt with a one-frame limit, built from fixed, known-valid values
// (ten prefill positions). Rows of the batcher's session before a ticket enters them; rows of a ragged first batch before prefill.
q3_request idle_request(int chunk_frames) {
    static const uint32_t one_tok[1] = {0};
    q3_request d{};
    d.mode = Q3_MODE_CUSTOM_VOICE; d.text_ids = one_tok; d.n_text = 1;
    d.speaker_id = 0; d.language_id = 0;                                       // codec token 0: valid in every vocabulary
    d.opts.temperature = 0.9; d.opts.top_p = 0.9; d.opts.repetition_penalty = 1.05; d.opts.top_k = 50;      // SynthesisOptions::default (lib.rs:1786-1836)
    d.opts.eos_token_id = -1; d.opts.min_new_tokens = 2; d.opts.max_length = 1; d.opts.has_seed = 1; d.opts.seed = 0;
    d.opts.chunk_frames = chunk_frames;
    return d;
}
// Sessions take rows of ANY mix of prompt kinds and lengths (round 5; BASELINE config[3] on a Base checkpoint mixes x-vector and
// ICL prompts, lib.rs:718-784, 802-870, 897-1046 are per-call in the reference). Rows of one prefill length are prefilled
// together by the session itself (the fast path: one batched prefill). A RAGGED batch is opened on idle rows sized for the
// longest prompt / largest limit; q3_session_prefill then prefills the rows in groups of equal prefill length — each group a
// batched side prefill — and moves every row in the way a continuous-batching swap does (transplant_row), so every row carries
// the bits of its own run and decode proceeds in one captured frame graph over all rows.
static q3_status session_create_any(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out) {
    if (!m || !reqs || !out) return set_err(Q3_INVALID_ARG, "q3_session_create: null argument");
    if (batch < 1 || batch > Q3_MAX_BATCH) return set_err(Q3_UNSUPPORTED, "batch %d unsupported (1..%d sequences per session)", batch, Q3_MAX_BATCH);
    bool ragged = false; int S0 = 0, L0 = 0, Smax = 0, Lmax = 0, rows_max = 0;
    for (int b = 0; b < batch; ++b) {
        const q3_request& r = reqs[b];
        if (r.n_text < 0 || r.n_instruct < 0 || r.n_ref < 0 || r.n_ref_text < 0) return set_err(Q3_INVALID_ARG, "bad token id arrays");
        int S = 0, L = 0; request_shape(r, &S, &L);
        if (b == 0) { S0 = S; L0 = L; }
        ragged = ragged || S != S0;
        Smax = std::max(Smax, S); Lmax = std::max(Lmax, L);
        rows_max = std::max(rows_max, (r.mode == Q3_MODE_VOICE_DESIGN ? r.n_instruct : 0) + 5 + r.n_ref_text + r.n_text + 1);
        if (r.opts.chunk_frames != reqs[0].opts.chunk_frames) return set_err(Q3_UNSUPPORTED, "all requests of a batch must share chunk_frames (the streaming chunk is a property of the session)");
    }
    (void)L0;
    if (!ragged) return session_create(m, reqs, batch, frame_budget, prompt_budget, out);
    if (Lmax < 1) return set_err(Q3_INVALID_ARG, "max_length must be >= 1");
    const int fb = std::max(frame_budget, Lmax);
    int pb = std::max(std::max(prompt_budget, Smax), 16);
    if (rows_max > pb + 1024) pb = rows_max - 1024;          // a row's text-row slot is prompt_budget + 1024 rows
    std::vector<q3_request> idle((size_t)batch, idle_request(reqs[0].opts.chunk_frames));
    q3_session* s = nullptr;
    Q3C(session_create(m, idle.data(), batch, fb, pb, &s));
    s->ragged.resize((size_t)batch);
    for (int b = 0; b < batch; ++b) s->ragged[(size_t)b].own(reqs[b], m->cfg.hidden);
    *out = s;
    return Q3_OK;
}
extern "C" q3_status q3_session_create(q3_model* m, const q3_request* reqs, int batch, q3_session** out) {
    return session_create_any(m, reqs, batch, 0, 0, out);
}
extern "C" q3_status q3_session_create_reserved(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out) {
    if (frame_budget < 0 || prompt_budget < 0) return set_err(Q3_INVALID_ARG, "q3_session_create_reserved: negative budget");
    return session_create_any(m, reqs, batch, frame_budget, prompt_budget, out);
}
q3_status session_create(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out,
                                hipStream_t borrow) {
    if (!m || !reqs || !out) return set_err(Q3_INVALID_ARG, "q3_session_create: null argument");
    if (!m->finalized) return set_err(Q3_INVALID_ARG, "model not finalized");
    if (batch < 1 || batch > Q3_MAX_BATCH) return set_err(Q3_UNSUPPORTED, "batch %d unsupported (1..%d sequences per session)", batch, Q3_MAX_BATCH);
    HIPC(hipSetDevice(m->device));
    const q3_config& c = m->cfg;
    std::unique_ptr<q3_session> s(new q3_session());
    s->m = m; s->B = batch; s->opts = reqs[0].opts;
    m->refs.fetch_add(1);
    if (s->opts.max_length < 1) return set_err(Q3_INVALID_ARG, "max_length must be >= 1");
    s->seq.resize(batch);
    int rows = 0;
    for (int b = 0; b < batch; ++b) {
        const q3_request& r = reqs[b];
        const bool icl_req = r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes && r.ref_text_ids;
        if (r.opts.chunk_frames != s->opts.chunk_frames) return set_err(Q3_UNSUPPORTED, "all requests of a batch must share chunk_frames (the streaming chunk is a property of the session)");
        if (!(r.opts.temperature >= 0.0) || r.opts.repetition_penalty <= 0.0) return set_err(Q3_INVALID_ARG, "bad sampling options");
        if (r.mode < 0 || r.mode > 2) return set_err(Q3_INVALID_ARG, "bad mode %d", r.mode);
        if (r.n_text < 0 || r.n_instruct < 0 || (r.n_text > 0 && !r.text_ids) || (r.n_instruct > 0 && !r.instruct_ids))
            return set_err(Q3_INVALID_ARG, "bad token id arrays");
        if (r.mode == Q3_MODE_VOICE_CLONE && !r.xvector) return set_err(Q3_INVALID_ARG, "voice clone needs an x-vector");
        SeqInfo& q = s->seq[b];
        q.req = r;
        q.text.assign(r.text_ids, r.text_ids + r.n_text);
        for (uint32_t id : q.text) if (id >= (uint32_t)c.text_vocab) return set_err(Q3_INVALID_ARG, "text id %u out of range", id);
        if (r.mode == Q3_MODE_VOICE_DESIGN) q.instruct.assign(r.instruct_ids, r.instruct_ids + r.n_instruct);
        for (uint32_t id : q.instruct) if (id >= (uint32_t)c.text_vocab) return set_err(Q3_INVALID_ARG, "instruct id %u out of range", id);
        if (r.language_id >= (uint32_t)c.codec_vocab || (r.mode == Q3_MODE_CUSTOM_VOICE && r.speaker_id >= (uint32_t)c.codec_vocab))
            return set_err(Q3_INVALID_ARG, "speaker/language id out of range");
        if (r.xvector) q.xvec.assign(r.xvector, r.xvector + c.hidden);
        q.icl = r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes && r.ref_text_ids;
        if (r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes) {
            // reference frames are prepended at decode whenever the prompt carries them — also without a reference
            // transcript, when the prefill stays x-vector-only (lib.rs:1022: `if let Some(ref_codes) = &prompt.ref_codes`)
            q.ref_codes.assign(r.ref_codes, r.ref_codes + (size_t)r.n_ref * 16);
            for (int f = 0; f < r.n_ref; ++f) {
                if (q.ref_codes[(size_t)f * 16] >= (uint32_t)c.codec_vocab) return set_err(Q3_INVALID_ARG, "reference semantic code out of range");
                for (int g = 1; g < 16; ++g) if (q.ref_codes[(size_t)f * 16 + g] >= (uint32_t)c.cp_vocab) return set_err(Q3_INVALID_ARG, "reference acoustic code out of range");
            }
        }
        if (q.icl) {
            if (r.n_ref_text < 0) return set_err(Q3_INVALID_ARG, "bad reference text");
            q.ref_text.assign(r.ref_text_ids, r.ref_text_ids + r.n_ref_text);
            for (uint32_t id : q.ref_text) if (id >= (uint32_t)c.text_vocab) return set_err(Q3_INVALID_ARG, "reference text id %u out of range", id);
            // lib.rs:913-929 (must be identical for every sequence of the batch, checked below through s->opts)
            q.req.opts.repetition_penalty = r.opts.repetition_penalty < 1.5 ? 1.5 : r.opts.repetition_penalty;
            int cap = 6 * r.n_text; if (cap < 75) cap = 75;
            if (q.req.opts.max_length > cap) q.req.opts.max_length = cap;
            if (b == 0) s->opts = q.req.opts;
        }
        const int n_ins = (int)q.instruct.size();
        const int overlay = r.mode == Q3_MODE_VOICE_DESIGN ? 5 : 6;
        const int n_icl = q.icl ? r.n_ref + 1 : 0;                              // streaming overlay: icl_len = n_codec
        const int n_text_all = q.icl ? r.n_ref_text + r.n_text + 1 : 0;         // [ref_text, text, tts_eos]
        q.prefill_len = n_ins + 3 + overlay + ((r.n_text > 0 && !q.icl) ? 1 : 0) + n_icl;
        q.trailing_len = q.icl ? (n_text_all > n_icl ? n_text_all - n_icl : 1) : (r.n_text > 1 ? r.n_text - 1 : 0) + 1;
        q.row_base = rows;
        q.n_rows = n_ins + 5 + (q.icl ? r.n_ref_text : 0) + r.n_text + 1;     // instruct, role×3, pad, bos, [ref_text…], text…, eos
        rows += q.n_rows;
        if (q.prefill_len != s->seq[0].prefill_len)
            return set_err(Q3_UNSUPPORTED, "all sequences of a batch must have the same prefill length (%d vs %d)", q.prefill_len, s->seq[0].prefill_len);
    }
    s->prefill_len = s->seq[0].prefill_len;
    s->n_rows_total = rows;
    s->max_frames = frame_budget > 1 ? frame_budget : 1;          // room for rows that arrive later with a larger limit (continuous batching)
    for (auto& q : s->seq) {
        if (q.req.opts.max_length < 1) return set_err(Q3_INVALID_ARG, "max_length must be at least 1");
        q.limit = q.req.opts.max_length; q.start_run = 0;
        if (q.limit > s->max_frames) s->max_frames = q.limit;
        if (q.n_rows > s->row_cap) s->row_cap = q.n_rows;
    }
    s->opts.max_length = s->max_frames;
    if (s->row_cap < 1024) s->row_cap = 1024;      // replacement slots hold any text up to ~1000 tokens (8 MB per row at H = 2048), longer if the batch had one
    if (prompt_budget > 0 && s->row_cap < prompt_budget + 1024) s->row_cap = prompt_budget + 1024;      // ... or the caller announced longer prompts (instruct / reference text rows are projected rows too)
    s->repl_base = rows;
    // KV sized for what the path needs (prefill + frames), not the reference's max_new_tokens+256 (lib.rs:450)
    s->max_seq = (prompt_budget > s->prefill_len ? prompt_budget : s->prefill_len) + s->max_frames + 1;   // prompt_budget: later rows with longer prompts
    if (s->max_seq > m->rope_len) return set_err(Q3_KV_OVERFLOW, "sequence length %d exceeds the RoPE table (%d)", s->max_seq, m->rope_len);
    {
        static const int ns_env = [] { const char* e = getenv("Q3_ATTN_SPLITS"); return e ? atoi(e) : 0; }();   // tuning aid
        // ~2 attention workgroups per CU over a 640-frame utterance. With the three-deep unconditional K/V requests of
        // k_attn_fused (round 2) a workgroup walks its keys without a round trip per key and fewer, longer key ranges
        // win: B = 8: 3.908 / 3.724 / 3.651 / 3.609 / 3.635 / 3.812 ms/frame at 1 / 2 / 4 / 8 / 16 / 32 splits (before:
        // 4.40 / 3.99 / 3.77 / 3.67 / 3.68); B = 1 stays at 16 (2.794 vs 2.800 at 8)
        int ns = ns_env > 0 ? ns_env : (batch <= 2 ? 1024 : 512) / (batch * c.n_kv_heads);     // (B <= 2: up to the long-context cap below)
        // long contexts (a 4k-position prompt: 38 MB of f32 K/V per layer) want more than 16 workgroups per KV head to
        // stream them (B = 1 at 4.1k positions: 3.90 -> 3.55 ms/frame); short sessions keep the cheaper 16-way merge
        const int cap = ns_env > 0 ? MAX_SPLITS : (s->max_seq > 2048 ? MAX_SPLITS : 16);
        if (ns < 1) ns = 1; if (ns > cap) ns = cap; s->n_splits = ns;
    }
    {   // the frame loop is a chain of ~600 short dependent kernels per frame: give its queue the highest priority so
        // that its workgroups are dispatched ahead of the vocoder segments running beside it (q3_session_run)
        if (borrow) { s->stream = borrow; s->owns_stream = false; }      // q3_session_replace: no second queue for a one-row prefill
        else {
            {
                std::lock_guard<std::mutex> g(m->stream_mu);
                if (!m->idle_streams.empty()) { s->stream = m->idle_streams.back(); m->idle_streams.pop_back(); }
            }
            if (!s->stream) {
                int least = 0, greatest = 0;
                HIPC(hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIPC(hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, greatest));
            }
        }
    }
    const int B = batch, H = c.hidden, CH = c.cp_hidden;
    s->pool.lazy = true;
    auto alloc_lm = [&](LmBuf& b, const LmDims& d, int nsplit) -> hipError_t {
        const int QD = d.nh * HEAD_DIM, KD = d.nkv * HEAD_DIM;
        const size_t R = batch > 16 ? (size_t)up16(batch) : 16;     // rows: up to 16 for multi-row steps (B <= 16), else one row per sequence
        hipError_t e;
        if ((e = s->pool.alloc(&b.X, R * d.H)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.SUM, R * d.H)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.QKV, R * (QD + 2 * KD))) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.Q, R * QD)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.ATT, R * QD)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.ACT, R * d.I)) != hipSuccess) return e;
        return s->pool.alloc(&b.PART, R * d.nh * nsplit * PART_STRIDE);
    };
    HIPC(alloc_lm(s->tb, talker_dims(c), s->n_splits));
    HIPC(alloc_lm(s->cb, cp_dims(c), 1));
    if (B > 16) {     // wide sessions: workspace of the split-K GEMM, sized for the largest projection of either network
        size_t need = 0;
        auto upd = [&](int N, int K, int epi) { if (N % 128 == 0 && K % 128 == 0) { const size_t b = gemm_wide_ws_bytes(B, N, K, epi); if (b > need) need = b; } };
        const int QDt = c.n_heads * HEAD_DIM, KDt = c.n_kv_heads * HEAD_DIM, QDc = c.cp_heads * HEAD_DIM, KDc = c.cp_kv_heads * HEAD_DIM;
        upd(QDt + 2 * KDt, H, EPI_NONE); upd(H, QDt, EPI_RESID); upd(c.inter, H, EPI_SWIGLU); upd(H, c.inter, EPI_RESID); upd(c.codec_vocab, H, EPI_NONE);
        upd(QDc + 2 * KDc, c.cp_hidden, EPI_NONE); upd(c.cp_hidden, QDc, EPI_RESID); upd(c.cp_inter, c.cp_hidden, EPI_SWIGLU); upd(c.cp_hidden, c.cp_inter, EPI_RESID);
        upd(c.cp_vocab, c.cp_hidden, EPI_NONE); upd(c.cp_hidden, H, EPI_NONE);
        if (need) { HIPC(s->pool.alloc(&s->wide_ws, need / 4)); s->wide_ws_bytes = need; }
    }
    HIPC(s->pool.alloc(&s->LASTH, (size_t)B * H));
    HIPC(s->pool.alloc(&s->LOGITS, (size_t)B * c.codec_vocab));
    HIPC(s->pool.alloc(&s->CP_IN, (size_t)(B > 16 ? up16(B) : 16) * H));
    HIPC(s->pool.alloc(&s->CP_LOGITS, (size_t)15 * B * c.cp_vocab));
    s->kv_layer_stride = (size_t)B * c.n_kv_heads * s->max_seq * HEAD_DIM;
    {   // talker KV: pages from the model's pool as the rows grow (default), or one extent per row sized for the worst case
        const char* e = getenv("Q3_KV_CONTIGUOUS");          // read per session: the paged-vs-contiguous test flips it
        s->paged = !(e && atoi(e) != 0);
        if (s->paged) {
            if (s->max_seq > KV_MAX_PAGES * KV_PAGE_POS) return set_err(Q3_KV_OVERFLOW, "sequence length %d exceeds a row's page table (%d)", s->max_seq, KV_MAX_PAGES * KV_PAGE_POS);
            HIPC(s->pool.alloc(&s->kv_table, (size_t)B * KV_MAX_PAGES));
            HIPC(s->pool.alloc(&s->kv_conv, (size_t)2 * B * KV_MAX_PAGES));
            s->kv_rows.resize((size_t)B);
            for (auto& r : s->kv_rows) r.reserve(KV_MAX_PAGES);
        } else {
            HIPC(s->pool.alloc(&s->kcache, s->kv_layer_stride * c.n_layers));
            HIPC(s->pool.alloc(&s->vcache, s->kv_layer_stride * c.n_layers));
        }
    }
    s->ckv_layer_stride = (size_t)B * c.cp_kv_heads * (c.n_groups + 1) * HEAD_DIM;
    // K and V of the code predictor in ONE block: k_attn_cp takes their distance as a 32-bit float count, and two blocks of the
    // size-class cache can lie further apart than that (the launch then silently fell back to k_attn_fused: 70 nodes per
    // frame 1.1 us slower each — seen in a profiling run of round 4)
    HIPC(s->pool.alloc(&s->ckcache, 2 * s->ckv_layer_stride * c.cp_layers + 64));
    s->cvcache = s->ckcache + s->ckv_layer_stride * c.cp_layers + 64;
    HIPC(s->pool.alloc(&s->rows, ((size_t)rows + (size_t)B * s->row_cap) * H));      // + one replacement slot per row (q3_session_replace)
    HIPC(s->pool.alloc(&s->limit, B));
    HIPC(s->pool.alloc(&s->sample_rows, B));
    {
        std::vector<int> lim(B); std::vector<SampleRow> sr(B);
        for (int b = 0; b < B; ++b) { lim[b] = s->seq[b].limit; sr[b] = sample_row(s->seq[b].req.opts); }
        HIPC(q3_hipMemcpy(s->limit, lim.data(), B * 4, hipMemcpyHostToDevice));
        HIPC(q3_hipMemcpy(s->sample_rows, sr.data(), B * sizeof(SampleRow), hipMemcpyHostToDevice));
    }
    HIPC(s->pool.alloc(&s->embeds, (size_t)B * s->prefill_len * H));
    HIPC(s->pool.alloc(&s->xvec, (size_t)B * H));
    HIPC(s->pool.alloc(&s->ids_dev, (size_t)rows)); HIPC(s->pool.alloc(&s->tr_dev, (size_t)B * s->prefill_len)); HIPC(s->pool.alloc(&s->ci_dev, (size_t)B * s->prefill_len));
    HIPC(s->pool.alloc(&s->proj_e, (size_t)rows * c.text_dim)); HIPC(s->pool.alloc(&s->proj_h, (size_t)rows * c.text_dim));
    HIPC(s->pool.alloc(&s->trail_base, B)); HIPC(s->pool.alloc(&s->trail_len, B)); HIPC(s->pool.alloc(&s->pad_row, B));
    HIPC(s->pool.alloc(&s->tok, B)); HIPC(s->pool.alloc(&s->seen, (size_t)B * c.codec_vocab));
    HIPC(s->pool.alloc(&s->frame_idx, B)); HIPC(s->pool.alloc(&s->pos, B)); HIPC(s->pool.alloc(&s->token_count, B));
    HIPC(s->pool.alloc(&s->U, (size_t)B * (s->max_frames + 2)));      // one draw per sampled token (max_frames + 1) + a spare for a frozen row
    HIPC(s->pool.alloc(&s->codes, (size_t)B * s->max_frames * 16));
    // RNG: one PCG stream per sequence, one draw per sampled token
    std::vector<float> U((size_t)B * (s->max_frames + 2), 0.0f);
    for (int b = 0; b < B; ++b) {
        uint64_t st;
        const q3_options& o = reqs[b].opts;
        uint64_t seed = o.seed;
        if (!o.has_seed) seed = (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count() + 0x9E37ULL * b;
        q3_rng_seed(seed, &st);
        for (int i = 0; i <= s->max_frames; ++i) U[(size_t)b * (s->max_frames + 2) + i] = q3_rng_next(&st);
    }
    HIPC(q3_hipMemcpy(s->U, U.data(), U.size() * 4, hipMemcpyHostToDevice));
    HIPC(s->pool.settle());                 // every zero-fill has landed before a kernel on the session's own stream can run
    s->pool.lazy = false;
    *out = s.release();
    return Q3_OK;
}

// Every exit path (q3_session_free, a failed q3_session_create) goes through here: streams are drained BEFORE the pool's
// blocks return to the device cache (member destructors run after this body), so no later session is handed memory a
// queued kernel still writes.
q3_session::~q3_session() {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (dec_stream) (void)hipStreamSynchronize(dec_stream);
    for (auto st : par_streams) (void)hipStreamSynchronize(st);
    if (aql) q3::aql_program_destroy(aql);               // waits for its outstanding replays
    for (int b = 0; b < (int)kv_rows.size(); ++b) kv_release_row(this, b);      // every stream that touched them is idle
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    for (auto& ev : prof_pool) (void)hipEventDestroy(ev);
    for (auto st : par_streams) (void)hipStreamDestroy(st);
    cws.release(); seg_ws.release();
    for (auto& w : par_ws) w.release();
    if (dec_ev) (void)hipEventDestroy(dec_ev);
    if (dec_stream) (void)hipStreamDestroy(dec_stream);
    if (stream && owns_stream) {                   // synchronised above: idle, handed to the next session of this model
        std::lock_guard<std::mutex> g(m->stream_mu);
        if (m->idle_streams.size() < 16) { m->idle_streams.push_back(stream); stream = nullptr; }
    }
    if (stream && owns_stream) (void)hipStreamDestroy(stream);
    pool.release_all();                                          // the model's device must still be current for these
    if (m->refs.fetch_sub(1) == 1) model_destroy(m);          // the model handle was given up before its last session
}

extern "C" void q3_session_free(q3_session* s) { delete s; }

// K/V dtype of the talker cache (before q3_session_prefill): Q3_DTYPE_F32 (default: the parity contract is the reference's CPU
// F32 path) or Q3_DTYPE_BF16 — the reference GPU path's cache dtype (kv_cache.rs:234-310): half the K/V bytes per frame, results
// no longer bit-comparable with the F32 oracle. Needs the paged cache.
extern "C" q3_status q3_session_set_kv_dtype(q3_session* s, int dtype) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->prefilled) return set_err(Q3_INVALID_ARG, "q3_session_set_kv_dtype must be called before prefill");
    if (dtype != Q3_DTYPE_F32 && dtype != Q3_DTYPE_BF16) return set_err(Q3_INVALID_ARG, "q3_session_set_kv_dtype: unknown dtype %d", dtype);
    if (dtype == Q3_DTYPE_BF16 && !s->paged) return set_err(Q3_UNSUPPORTED, "bf16 K/V needs the paged cache (Q3_KV_CONTIGUOUS is set)");
    s->kv_bf16 = dtype == Q3_DTYPE_BF16;
    return Q3_OK;
}

extern "C" q3_status q3_session_set_debug(q3_session* s, int capture) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->prefilled) return set_err(Q3_INVALID_ARG, "set_debug must be called before prefill");
    s->debug = capture != 0;
    if (s->debug && !s->logits_hist) {
        const q3_config& c = s->m->cfg;
        HIPC(hipSetDevice(s->m->device));
        HIPC(s->pool.alloc(&s->logits_hist, (size_t)s->B * (s->max_frames + 1) * c.codec_vocab));
        HIPC(s->pool.alloc(&s->cp_logits_hist, (size_t)s->max_frames * 15 * s->B * c.cp_vocab));
    }
    return Q3_OK;
}
extern "C" q3_status q3_session_set_profile(q3_session* s, int enable) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    s->profile = enable != 0;
    return Q3_OK;
}
extern "C" q3_status q3_session_stream(q3_session* s, void** stream) {
    if (!s || !stream) return set_err(Q3_INVALID_ARG, "null argument");
    *stream = (void*)s->stream;
    return Q3_OK;
}
extern "C" q3_status q3_session_prefill_len(q3_session* s, int b, int* prefill_len, int* trailing_len) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    if (!s->ragged.empty()) {                    // a ragged batch before its prefill: the rows are still idle placeholders
        int S = 0, L = 0; request_shape(s->ragged[(size_t)b].r, &S, &L);
        if (prefill_len) *prefill_len = S;
        if (trailing_len) *trailing_len = -1;
        return Q3_OK;
    }
    if (prefill_len) *prefill_len = s->seq[b].prefill_len;
    if (trailing_len) *trailing_len = s->seq[b].trailing_len;
    return Q3_OK;
}

// text projection (talker.rs:316-320) of `n` gathered rows: fc2(silu(fc1(e)+b1))+b2, 8 rows a time
static q3_status text_project(q3_session* s, const uint32_t* ids_dev, int n, float* out_rows) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const int TD = c.text_dim, H = c.hidden;
    if (n > s->n_rows_total) return set_err(Q3_INVALID_ARG, "text_project: %d rows exceed the session's %d", n, s->n_rows_total);
    float *e = s->proj_e, *h = s->proj_h;      // session scratch: everything below is queued on s->stream, nothing waits
    q3_status st = Q3_OK;
    hipError_t er = launch_gather_rows_bf16(m->text_emb, ids_dev, e, n, TD, s->stream);
    if (er == hipSuccess && n >= 48 && TD % 64 == 0 && H % 64 == 0 && !s->no_chunk) {
        // TextProjection (talker.rs:316-320) over all n token rows as two GEMMs instead of n/8 GEMV passes
        GemmArgs g1; g1.W = m->fc1w.t1; g1.x = e; g1.ldx = TD; g1.bias = m->fc1b; g1.y = h; g1.ldy = TD;
        g1.M = n; g1.N = TD; g1.K = TD; g1.Kpad = (TD + 31) / 32 * 32; g1.epi = EPI_SILU;
        er = launch_lm_gemm(g1, s->stream);
        if (er == hipSuccess) {
            GemmArgs g2; g2.W = m->fc2w.t1; g2.x = h; g2.ldx = TD; g2.bias = m->fc2b; g2.y = out_rows; g2.ldy = H;
            g2.M = n; g2.N = H; g2.K = TD; g2.Kpad = (TD + 31) / 32 * 32; g2.epi = EPI_NONE;
            er = launch_lm_gemm(g2, s->stream);
        }
    } else
    for (int r0 = 0; r0 < n && er == hipSuccess; r0 += 8) {
        const int M = (n - r0) < 8 ? (n - r0) : 8;
        LinArgs a;
        a.N = TD; a.K = TD; set_w(a, m->fc1w, M, TD, TD); a.x = e + (size_t)r0 * TD; a.ldx = TD; a.bias = m->fc1b; a.y = h + (size_t)r0 * TD; a.ldy = TD; a.M = M; a.epi = EPI_SILU;
        er = launch_linear(a, s->stream);
        if (er != hipSuccess) break;
        LinArgs b2;
        b2.N = H; b2.K = TD; set_w(b2, m->fc2w, M, H, TD); b2.x = h + (size_t)r0 * TD; b2.ldx = TD; b2.bias = m->fc2b; b2.y = out_rows + (size_t)r0 * H; b2.ldy = H; b2.M = M; b2.epi = EPI_NONE;
        er = launch_linear(b2, s->stream);
    }
    if (er != hipSuccess) st = set_err(Q3_HIP_ERROR, "text projection: %s", hipGetErrorString(er));
    return st;
}

// run_prefill_layers (talker.rs:823-841) for long prompts: chunks of up to 128 positions per sequence go through every
// layer as GEMMs over the decode path's tiled weight image + a query-blocked causal attention (q3_kernels_prefill.hip).
// Leaves the KV cache filled for positions [0, S) and LASTH / LOGITS of the last position, like the chunked decode-step
// schedule it replaces for S >= 48.
static q3_status prefill_gemm(q3_session* s, int S_all, int S, bool with_head) {      // positions [0, S) of the S_all-position prompts
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const LmDims d = talker_dims(c);
    const int B = s->B, H = d.H, QD = d.nh * HEAD_DIM, KD = d.nkv * HEAD_DIM, I = d.I;
    // positions per sequence per pass: up to 4224 activation rows per launch (a whole 4k-position prompt in one pass:
    // 33 M-tiles x N/128 workgroups per GEMM, 520 attention workgroups; 2048-row passes: 113 ms instead of 99 for the
    // 4105-position prefill of the 1.7B model), at least one 128-row tile per sequence
    static const int rows_env = [] { const char* e = getenv("Q3_PREFILL_ROWS"); return e ? atoi(e) : 4224; }();
    int C = rows_env / B; C = C < 128 ? 128 : (C / 128) * 128;
    const int max_rows = B * (S < C ? S : C);
    struct SyncedPool : DevPool { hipStream_t st; explicit SyncedPool(hipStream_t s_) : st(s_) {} ~SyncedPool() { (void)hipStreamSynchronize(st); } };
    SyncedPool tmp(s->stream);       // on EVERY return path the stream is drained before the blocks go back to the cache
    float *X, *QKV, *Qb, *ATT, *SUM, *ACT, *DEN;
    HIPC(tmp.alloc(&X, (size_t)max_rows * H)); HIPC(tmp.alloc(&QKV, (size_t)max_rows * (QD + 2 * KD)));
    HIPC(tmp.alloc(&Qb, (size_t)max_rows * QD)); HIPC(tmp.alloc(&ATT, (size_t)max_rows * QD));
    HIPC(tmp.alloc(&SUM, (size_t)max_rows * H)); HIPC(tmp.alloc(&ACT, (size_t)max_rows * I)); HIPC(tmp.alloc(&DEN, (size_t)max_rows));
    auto kp = [](int K) { return (K + 31) / 32 * 32; };
    // every GEMM input is split once into its three exact bf16 terms (launch_split_rows) instead of once per workgroup
    // column inside the GEMM; Q3_GEMM_NO_PLANES=1 keeps the in-kernel split (A/B aid)
    // long prompts: every query block's key range is halved over two workgroups and merged (k_attn_prefill_t)
    static const bool no_split = getenv("Q3_PREFILL_ATTN_NOSPLIT") != nullptr;
    static const bool gen2_attn = getenv("Q3_PREFILL_ATTN_GEN2") != nullptr || getenv("Q3_PREFILL_ATTN_VALU") != nullptr;
    const bool kv_split = !no_split && !gen2_attn && S >= 1024;
    float* PART = nullptr;
    if (kv_split) HIPC(tmp.alloc(&PART, (size_t)max_rows * d.nh * 2 * PART_STRIDE));
    // bf16-matrix-core attention (k_attn_prefill_x3) over per-layer bf16x3 planes of the cached K/V; Q3_PREFILL_ATTN_X3=0
    // keeps the f32-MFMA generation (A/B aid, read per call)
    const char* x3e = getenv("Q3_PREFILL_ATTN_X3");
    const bool attn_x3 = !gen2_attn && !(x3e && atoi(x3e) == 0) && S >= 256;
    unsigned char* KVP = nullptr;
    const int kvp_tiles = (S + 31) / 32;
    if (attn_x3) HIPC(tmp.alloc(&KVP, (size_t)B * d.nkv * kvp_tiles * KVP_TILE_BYTES));
    static const bool no_planes = getenv("Q3_GEMM_NO_PLANES") != nullptr;
    const bool planes = !no_planes && H % 8 == 0 && QD % 8 == 0 && I % 8 == 0;
    const int kmax = std::max(kp(H), std::max(kp(QD), kp(I)));
    const size_t plane_elems = (size_t)((max_rows + 127) / 128 * 128) * kmax;      // whole 128-row tiles (GemmArgs::xp)
    uint16_t* XP = nullptr;
    if (planes) HIPC(tmp.alloc(&XP, plane_elems * 3));
    auto split = [&](GemmArgs& g) -> hipError_t {
        if (!planes) return hipSuccess;
        g.xp = XP; g.xp_plane = plane_elems;
        return launch_split_rows(g.x, g.ldx, g.norm_w, XP, plane_elems, g.M, g.K, g.Kpad, s->stream);
    };
    // split-K workspace of the third GEMM geometry: small row counts only (that is where a GEMM's grid underfills the chip)
    float* SKW = nullptr; size_t skw_bytes = 0;
    if (planes && max_rows <= 1024) { skw_bytes = (size_t)8 * max_rows * (QD + 2 * KD > 2 * I / 4 ? QD + 2 * KD : 2 * I / 4) * sizeof(float); HIPC(tmp.alloc(&SKW, skw_bytes / sizeof(float))); }
    int ch = 0;
    for (int t0 = 0; t0 < S; t0 += C) {
        ch = (S - t0) < C ? (S - t0) : C;
        const int rows = B * ch;
        for (int b = 0; b < B; ++b)
            HIPC(launch_copy_rows(s->embeds + ((size_t)b * S_all + t0) * H, H, X + (size_t)b * ch * H, H, ch, H, s->stream));
        for (int i = 0; i < d.layers; ++i) {
            const LayerW& w = m->tl[i];
            HIPC(launch_row_den(X, H, DEN, rows, H, d.eps, s->stream));
            GemmArgs g; g.W = w.qkv.t1; g.x = X; g.ldx = H; g.norm_w = w.in_ln; g.den = DEN; g.y = QKV; g.ldy = QD + 2 * KD;
            g.M = rows; g.N = QD + 2 * KD; g.K = H; g.Kpad = kp(H); g.epi = EPI_NONE;
            g.splitk_ws = SKW; g.splitk_ws_bytes = skw_bytes;
            HIPC(split(g)); HIPC(launch_lm_gemm(g, s->stream));
            AttnArgs t{};
            t.qkv = QKV; t.ld_qkv = QD + 2 * KD; t.q_norm_w = w.q_norm; t.k_norm_w = w.k_norm; t.eps = d.eps;
            t.rope_cos = m->rope_cos; t.rope_sin = m->rope_sin; t.pos_dev = nullptr; t.pos_static = t0;
            if (s->paged) {
                t.kv_pages = s->kv_table; t.kv_layer_off = (size_t)i * m->kv_pool.layer_stride(); t.kv_vdelta = m->kv_pool.v_delta();
            } else { t.kcache = s->kcache + (size_t)i * s->kv_layer_stride; t.vcache = s->vcache + (size_t)i * s->kv_layer_stride; }
            t.max_seq = s->max_seq; t.qbuf = Qb; t.part = nullptr; t.out = ATT; t.ld_out = QD;
            t.B = rows; t.nh = d.nh; t.nkv = d.nkv; t.n_splits = 1; t.rows_per_seq = ch;
            HIPC(launch_qknorm_rope_kv(t, s->stream));
            if (attn_x3) {
                HIPC(launch_kv_planes(t, B * d.nkv, t0 + ch, kvp_tiles, KVP, s->stream));
                t.kvp = KVP; t.kvp_tiles = kvp_tiles;
            }
            if (kv_split) { t.part = PART; t.n_splits = 2; }
            HIPC(launch_attn_prefill(t, s->stream));
            if (kv_split) HIPC(launch_attn_merge(t, s->stream));
            GemmArgs o; o.W = w.o.t1; o.x = ATT; o.ldx = QD; o.resid = X; o.ldr = H; o.y = SUM; o.ldy = H;
            o.M = rows; o.N = H; o.K = QD; o.Kpad = kp(QD); o.epi = EPI_RESID;
            o.splitk_ws = SKW; o.splitk_ws_bytes = skw_bytes;
            HIPC(split(o)); HIPC(launch_lm_gemm(o, s->stream));
            HIPC(launch_row_den(SUM, H, DEN, rows, H, d.eps, s->stream));
            GemmArgs gu; gu.W = w.gate.t1; gu.W2 = w.up.t1; gu.x = SUM; gu.ldx = H; gu.norm_w = w.post_ln; gu.den = DEN; gu.y = ACT; gu.ldy = I;
            gu.M = rows; gu.N = I; gu.K = H; gu.Kpad = kp(H); gu.epi = EPI_SWIGLU;
            gu.splitk_ws = SKW; gu.splitk_ws_bytes = skw_bytes;
            HIPC(split(gu)); HIPC(launch_lm_gemm(gu, s->stream));
            GemmArgs dn; dn.W = w.down.t1; dn.x = ACT; dn.ldx = I; dn.resid = SUM; dn.ldr = H; dn.y = X; dn.ldy = H;
            dn.M = rows; dn.N = H; dn.K = I; dn.Kpad = kp(I); dn.epi = EPI_RESID;
            dn.splitk_ws = SKW; dn.splitk_ws_bytes = skw_bytes;
            HIPC(split(dn)); HIPC(launch_lm_gemm(dn, s->stream));
        }
    }
    if (!with_head) { HIPC(sync_frames(s)); return Q3_OK; }      // the caller runs the remaining positions
    // head on each sequence's last position (row b*ch + ch-1 of the last chunk): final norm -> LASTH, codec_head -> LOGITS
    HIPC(launch_rmsnorm(X + (size_t)(ch - 1) * H, ch * H, m->norm, s->LASTH, H, B, H, c.rms_eps, s->stream));
    LinArgs h;
    h.N = c.codec_vocab; h.K = H; set_w(h, m->codec_head, B, h.N, h.K); h.x = s->LASTH; h.ldx = H; h.y = s->LOGITS; h.ldy = c.codec_vocab;
    h.M = B; h.epi = EPI_NONE;
    HIPC(run_linear(s, h));
    HIPC(sync_frames(s));      // tmp buffers are freed on return
    return Q3_OK;
}

// q3_session_prefill of a ragged first batch (session_create_any): the idle rows never run — every row's state comes from a side
// session. Rows are grouped by prefill length in row order; a group of G rows is one batched side prefill.
static q3_status prefill_ragged(q3_session* s) {
    if (s->debug || s->profile) return set_err(Q3_UNSUPPORTED, "debug / profiling sessions need rows of one prefill length");
    const int B = s->B;
    s->kv_in_bf16 = s->kv_bf16;                          // the idle rows hold no pages: nothing to convert
    s->prefilled = true; s->frames_run = 0; s->codes_host_valid = false;      // transplant_row stamps rows with start_run = frames_run
    std::vector<char> placed((size_t)B, 0);
    for (int b0 = 0; b0 < B; ++b0) {
        if (placed[(size_t)b0]) continue;
        int S0 = 0, L = 0; request_shape(s->ragged[(size_t)b0].r, &S0, &L);
        std::vector<int> rows; std::vector<q3_request> reqs; std::vector<int> limits;
        for (int b = b0; b < B; ++b) {
            int S = 0, Lb = 0; request_shape(s->ragged[(size_t)b].r, &S, &Lb);
            if (placed[(size_t)b] || S != S0) continue;
            q3_request r = s->ragged[(size_t)b].r;
            // the row's RESOLVED limit (an ICL row's max_length is capped at max(75, 6 * n_text), talker.rs:646-710 / lib.rs:897-1046) is
            // what session_create_any sized max_frames for — the raw max_length of an ICL row may well exceed it
            if (r.opts.max_length < 1 || Lb < 1 || Lb > s->max_frames) { s->prefilled = false; return set_err(Q3_INVALID_ARG, "row %d: max_length %d (resolved %d) outside 1..%d", b, r.opts.max_length, Lb, s->max_frames); }
            limits.push_back(Lb);
            r.opts.max_length = s->max_frames;           // the side session draws the row's PCG stream with the host session's stride
            rows.push_back(b); reqs.push_back(r); placed[(size_t)b] = 1;
        }
        q3_session* side_raw = nullptr;
        q3_status st = session_create(s->m, reqs.data(), (int)reqs.size(), 0, 0, &side_raw, s->stream);
        std::unique_ptr<q3_session> side(side_raw);
        if (st == Q3_OK) { side->kv_bf16 = s->kv_bf16; }
        std::vector<int> lim(rows.size(), 0);
        for (size_t j = 0; j < rows.size() && st == Q3_OK; ++j) st = transplant_check(s, side.get(), (int)j, limits[j], &lim[j]);
        if (st == Q3_OK) st = q3_session_prefill(side.get());
        if (st == Q3_OK && sync_frames(s) != hipSuccess) st = set_err(Q3_HIP_ERROR, "ragged prefill: stream");
        for (size_t j = 0; j < rows.size() && st == Q3_OK; ++j) st = transplant_row(s, rows[j], side.get(), (int)j, lim[j]);
        if (st != Q3_OK) { s->prefilled = false; return st; }
    }
    s->ragged.clear();
    return Q3_OK;
}

static q3_status frame_capture(q3_session* s, bool stream_busy);
extern "C" q3_status q3_session_prefill(q3_session* s) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->prefilled) return set_err(Q3_INVALID_ARG, "session already prefilled");
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    HIPC(hipSetDevice(m->device));
    if (!s->ragged.empty()) return prefill_ragged(s);
    const int B = s->B, H = c.hidden, S = s->prefill_len;
    for (int b = 0; b < B; ++b) Q3C(kv_reserve_row(s, b, S + 1));      // paged KV: the prompt's positions and the first frame's
    // 1. ids to project, per sequence: [instruct…, IM_START, ASSISTANT, NEWLINE, TTS_PAD, TTS_BOS, text…, TTS_EOS]
    std::vector<uint32_t> ids; ids.reserve(s->n_rows_total);
    std::vector<int> text_row((size_t)B * S, -1), codec_id((size_t)B * S, -1);
    std::vector<int> trail_base(B), trail_len(B), pad_row(B);
    std::vector<float> xv((size_t)B * H, 0.0f);
    for (int b = 0; b < B; ++b) {
        SeqInfo& q = s->seq[b];
        const int n_ins = (int)q.instruct.size(), n_text = (int)q.text.size(), base = q.row_base;
        for (uint32_t id : q.instruct) ids.push_back(id);
        ids.push_back(IM_START); ids.push_back(ASSISTANT); ids.push_back(NEWLINE); ids.push_back(TTS_PAD); ids.push_back(TTS_BOS);
        if (q.icl) for (uint32_t id : q.ref_text) ids.push_back(id);       // ICL: [ref_text, target_text, tts_eos] (talker.rs:656-663)
        for (uint32_t id : q.text) ids.push_back(id);
        ids.push_back(TTS_EOS);
        const int n_ref_text = q.icl ? (int)q.ref_text.size() : 0;
        const int r_role = base + n_ins, r_pad = r_role + 3, r_bos = r_pad + 1, r_text = r_bos + 1, r_eos = r_text + n_ref_text + n_text;
        const int n_icl = q.icl ? (int)(q.ref_codes.size() / 16) + 1 : 0, n_text_all = n_ref_text + n_text + 1;
        q.pad_row = r_pad;
        if (q.icl) q.trail_base = n_text_all > n_icl ? r_text + n_icl : r_pad;       // talker.rs:692-708
        else q.trail_base = n_text > 1 ? r_text + 1 : r_eos;          // build_trailing_text (lib.rs:508-519)
        trail_base[b] = q.trail_base; trail_len[b] = q.trailing_len; pad_row[b] = q.pad_row;
        // prefill positions (talker.rs:451-491 / 511-564 / 585-627)
        int* tr = &text_row[(size_t)b * S]; int* ci = &codec_id[(size_t)b * S];
        int p = 0;
        for (int i = 0; i < n_ins; ++i) tr[p++] = base + i;
        for (int i = 0; i < 3; ++i) tr[p++] = r_role + i;
        const bool vd = q.req.mode == Q3_MODE_VOICE_DESIGN;
        int codec[7]; int nc;
        if (vd) { int t[6] = {CODEC_THINK, CODEC_THINK_BOS, (int)q.req.language_id, CODEC_THINK_EOS, CODEC_PAD, CODEC_BOS}; memcpy(codec, t, sizeof t); nc = 6; }
        else { int t[7] = {CODEC_THINK, CODEC_THINK_BOS, (int)q.req.language_id, CODEC_THINK_EOS, (int)q.req.speaker_id, CODEC_PAD, CODEC_BOS}; memcpy(codec, t, sizeof t); nc = 7; }
        const int overlay = nc - 1;
        for (int i = 0; i < overlay; ++i) {
            tr[p] = (i == overlay - 1) ? r_bos : r_pad;
            ci[p] = (q.req.mode == Q3_MODE_VOICE_CLONE && i == 4) ? -2 : codec[i];
            ++p;
        }
        if (n_text > 0 && !q.icl) { tr[p] = r_text; ci[p] = codec[nc - 1]; ++p; }
        for (int i = 0; i < n_icl; ++i) {      // ICL block: text (or tts_pad) row + codec_bos / Σ16 reference-frame embeddings
            tr[p] = i < n_text_all ? r_text + i : r_pad;
            ci[p] = i == 0 ? CODEC_BOS : -3 - (i - 1);
            ++p;
        }
        if (!q.xvec.empty()) memcpy(&xv[(size_t)b * H], q.xvec.data(), (size_t)H * 4);
    }
    uint32_t* ids_dev = s->ids_dev; int *tr_dev = s->tr_dev, *ci_dev = s->ci_dev;
    if ((int)ids.size() != s->n_rows_total) return set_err(Q3_INVALID_ARG, "prefill: row count mismatch (%zu vs %d)", ids.size(), s->n_rows_total);
    // reference frames of ICL sequences (also needed later by the ICL decode)
    std::vector<size_t> ref_off(B, 0); size_t ref_total = 0;
    for (int b = 0; b < B; ++b) { ref_off[b] = ref_total; ref_total += s->seq[b].ref_codes.size(); }
    if (ref_total && !s->ref_codes_dev) {
        HIPC(s->pool.alloc(&s->ref_codes_dev, ref_total));
        for (int b = 0; b < B; ++b)
            if (!s->seq[b].ref_codes.empty())
                HIPC(q3_hipMemcpy(s->ref_codes_dev + ref_off[b], s->seq[b].ref_codes.data(), s->seq[b].ref_codes.size() * 4, hipMemcpyHostToDevice));
    }
    // uploads ride the session stream (the host vectors live until the synchronisation that ends this function)
    HIPC(hipMemcpyAsync(ids_dev, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(tr_dev, text_row.data(), text_row.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(ci_dev, codec_id.data(), codec_id.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->xvec, xv.data(), xv.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->trail_base, trail_base.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->trail_len, trail_len.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->pad_row, pad_row.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    q3_status st = text_project(s, ids_dev, (int)ids.size(), s->rows);
    if (st == Q3_OK) {
        hipError_t e = hipSuccess;
        for (int b = 0; b < B && e == hipSuccess; ++b)
            e = launch_assemble_rows(s->rows, tr_dev + (size_t)b * S, m->codec_emb, ci_dev + (size_t)b * S, s->xvec + (size_t)b * H,
                                     s->embeds + (size_t)b * S * H, S, H, s->stream,
                                     s->ref_codes_dev ? s->ref_codes_dev + ref_off[b] : nullptr, m->cp_embs_dev);
        if (e != hipSuccess) st = set_err(Q3_HIP_ERROR, "prefill assembly: %s", hipGetErrorString(e));
    }
    if (st != Q3_OK) { (void)sync_frames(s); return st; }
    // 2. run_prefill_layers (talker.rs:823-841): causal attention ⇒ token-by-token decode steps
    //    and the GEMV kernels take up to 16 rows for the price of one, so each weight pass carries a CHUNK of
    //    16/B consecutive positions per sequence (q3_kernels.h AttnArgs::rows_per_seq). Bit-identical to the
    //    one-position-at-a-time schedule (rows are independent in the GEMV; attention sees the same K/V).
    static const int gemm_min = [] { const char* e = getenv("Q3_PREFILL_GEMM_MIN"); return e ? atoi(e) : 48; }();   // 0 disables the GEMM path
    const bool tiles_ok = (d_nh_ok(c));
    const int chunk = s->no_chunk ? 1 : (16 / B > 0 ? 16 / B : 1);
    // The GEMM path works in 128-position tiles and its grids are sized to fill the chip in whole rounds (4096
    // positions: 256 / 512 / 1536 workgroups of 256 CUs' worth); a few positions past the last full tile would cost every
    // GEMM another round (4105 positions, one sequence: +9 ... +50 % per GEMM). Up to Q3_PREFILL_TAIL (default 32)
    // trailing positions of a >= 1024-position prompt therefore go through the decode-step schedule below instead (about
    // 1 ms per 16 rows at 4k context), which appends to the same KV cache. The rule looks at the PROMPT only — never at the
    // batch — so which kernels compute a given position does not depend on how many sequences are prefilled together
    // (a batch pays ceil(B * r / 16) passes for it; positions below the cut always take the GEMM, whose bits are
    // batch-invariant; the decode-step kernels pick their tiling by the row count of a pass, like any decode step).
    static const int tail_max = [] { const char* e = getenv("Q3_PREFILL_TAIL"); return e ? atoi(e) : 32; }();
    int t_begin = 0;
    if (!s->no_chunk && !s->debug && gemm_min > 0 && S >= gemm_min && tiles_ok) {
        const int r = S % 128;
        const int Sg = (S >= 1024 && r > 0 && r <= tail_max) ? S - r : S;
        Q3C(prefill_gemm(s, S, Sg, Sg == S));
        t_begin = Sg;
    }
    for (int t0 = t_begin; t0 < S; t0 += chunk) {
        const int ch = (S - t0) < chunk ? (S - t0) : chunk;
        for (int b = 0; b < B; ++b)
            HIPC(launch_copy_rows(s->embeds + ((size_t)b * S + t0) * H, H, s->tb.X + (size_t)b * ch * H, H, ch, H, s->stream));
        Q3C(talker_step(s, nullptr, t0, t0 + ch >= S, ch));
    }
    // 3. first sampling decision (lib.rs:558-571)
    std::vector<int> posv(B, S), zero(B, 0);
    HIPC(hipMemcpyAsync(s->pos, posv.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->frame_idx, zero.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->token_count, zero.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    SampleArgs a; fill_sample_args(s, a); a.advance = 0;
    HIPC(launch_sample(a, s->stream));
    // Callers that are going to replay the frame (q3_session_run, q3_session_next_chunk) have it captured and converted HERE, while
    // the prompt's kernels run: ~2 ms of host work that used to sit between the prefill and the first frame (time to first audio).
    // (bf16-KV sessions capture later: which attention kernel the frame holds depends on the conversion below.)
    if (s->precapture && !s->kv_bf16 && !s->debug && !s->profile) Q3C(frame_capture(s, true));
    HIPC(sync_frames(s));
    if (s->kv_bf16 && !s->kv_in_bf16) Q3C(kv_convert_to_bf16(s));       // the prompt's K/V moves into pages of the bf16 pool, once
    s->prefilled = true; s->frames_run = 0; s->codes_host_valid = false;
    return Q3_OK;
}

static q3_status refresh_codes(q3_session* s) {
    if (s->codes_host_valid) return Q3_OK;
    HIPC(sync_frames(s));
    s->codes_host.resize((size_t)s->B * s->max_frames * 16);
    for (int b = 0; b < s->B; ++b) {
        int ran = s->frames_run - s->seq[b].start_run;
        if (ran > s->seq[b].limit) ran = s->seq[b].limit;
        if (ran > 0)
            HIPC(q3_hipMemcpy(&s->codes_host[(size_t)b * s->max_frames * 16], s->codes + (size_t)b * s->max_frames * 16,
                           (size_t)ran * 16 * 4, hipMemcpyDeviceToHost));
    }
    std::vector<uint32_t> tok(s->B);
    HIPC(q3_hipMemcpy(tok.data(), s->tok, s->B * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < s->B; ++b) {
        SeqInfo& q = s->seq[b];
        int n = s->frames_run - q.start_run; bool done = false;      // frames this row has run (rows swapped in later started later)
        if (n > q.limit) n = q.limit;
        if (n < 0) n = 0;
        const int eos = q.req.opts.eos_token_id;                     // per row (SampleRow)
        if (eos >= 0) {
            const int ran = n;
            for (int f = 0; f < ran; ++f)
                if ((int)s->codes_host[((size_t)b * s->max_frames + f) * 16] == eos) { n = f; done = true; break; }
            if (!done && ran < q.limit && (int)tok[b] == eos) done = true;     // EOS sampled for the next frame
        }
        if (n >= q.limit) done = true;
        q.n_frames = n; q.done = done;
    }
    s->codes_host_valid = true;
    return Q3_OK;
}

static bool all_done(q3_session* s) { for (auto& q : s->seq) if (!q.done) return false; return true; }
// frames the session still has to run for its longest-remaining row (lockstep sessions: max_frames - frames_run)
static int session_remaining(const q3_session* s) {
    int r = 0;
    for (const auto& q : s->seq) { const int left = q.limit - (s->frames_run - q.start_run); if (left > r) r = left; }
    return r;
}

// Which kernels of the frame keep to the activation-transport rule of q3_kernels.h (write-through + drained stores, L1-bypassing
// loads of everything an earlier node of the same frame wrote, nothing of it through the scalar cache): their packets go out
// without the agent-scope acquire / release fences (q3_aql.h). Anything not named here keeps HIP's fences — a kernel added to
// the frame later is safe by default. Development switches: Q3_AQL_T_ACQ=0 / Q3_AQL_T_REL=0 keep that half of every boundary,
// Q3_AQL_T_ONLY=<substring+substring> restricts the rule to kernels whose name holds one of the substrings (bisecting).
static void frame_fence_policy(const char* name, int* acquire, int* release) {
    static const char* const families[] = {"k_gemv_mfmaI", "k_gemv_sk2I", "k_gemv_gu24I", "k_gemv_ldsI", "k_gemv_mfma4I",
                                           "k_attn_cpI", "k_attn_fusedI", "k_attn_mergeI", "k_attn_first2I"};
    const bool drop_acq = !(getenv("Q3_AQL_T_ACQ") && atoi(getenv("Q3_AQL_T_ACQ")) == 0);
    const bool drop_rel = !(getenv("Q3_AQL_T_REL") && atoi(getenv("Q3_AQL_T_REL")) == 0);
    const std::string only = getenv("Q3_AQL_T_ONLY") ? getenv("Q3_AQL_T_ONLY") : "";
    bool conv = false;
    for (const char* f : families) conv = conv || strstr(name, f) != nullptr;
    if (conv && !only.empty()) {
        bool hit = false; size_t i = 0;
        while (i <= only.size()) {
            const size_t j = only.find_first_of(",+", i); const std::string t = only.substr(i, j == std::string::npos ? std::string::npos : j - i);
            if (!t.empty() && strstr(name, t.c_str())) hit = true;
            if (j == std::string::npos) break;
            i = j + 1;
        }
        conv = hit;
    }
    if (!conv) return;
    if (drop_acq) *acquire = 0;
    if (drop_rel) *release = 0;
}

// The frame is captured ONCE per session (frame_launch under stream capture: every per-frame quantity lives in device memory, so
// the graph is static) and turned into the packet program of the library's own AQL queue (q3_aql.cpp); only a graph the converter
// cannot take — or Q3_AQL=0 — is instantiated for hipGraphLaunch. `stream_busy`: the caller has work queued on the session's
// stream that must NOT be waited for here (q3_session_prefill captures while the prompt's kernels run: capture, conversion and
// the kernarg upload are host work, ~2 ms that used to sit between the prefill and the first frame of every session).
// Q3_AQL unset / 3 (round 6: the default; the whole -m gpu suite runs through it): boundaries between kernels that keep to the
// activation-transport rule (q3_kernels.h) without HIP's agent-scope fences. Q3_AQL=0: hipGraphLaunch; 1: own queue with HIP's
// fences on every packet (bit-identical to 0); 2: probe. Why a graph stayed on hipGraphLaunch: Q3_AQL_VERBOSE=1.
static q3_status frame_capture(q3_session* s, bool stream_busy) {
    // RELAXED capture mode: the captured region is nothing but kernel launches on the session's own non-blocking stream, while OTHER
    // host threads — the batcher's prefill worker, a server opening sessions on several threads — allocate, hipMemcpy and touch the
    // null stream as they please: in thread-local (or global) mode this HIP runtime fails THEIR calls ("operation not permitted
    // when stream is capturing") and invalidates the capture (tests/test_frame_submission.py: two sessions on two threads). An
    // invalidated capture is repeated all the same.
    for (int attempt = 0; !s->graph; ++attempt) {
        if (!stream_busy) HIPC(sync_frames(s));
        std::unique_lock<std::shared_mutex> cap(q3_capture_mu());      // no legacy-stream operation of any thread while the capture is open (q3_engine.h)
        {
            const hipError_t eb = hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed);
            if (eb != hipSuccess && stream_busy) { (void)hipGetLastError(); return Q3_OK; }      // not now: q3_session_generate captures on the idle stream
            HIPC(eb);
        }
        const q3_status st = frame_launch(s);
        const hipError_t e = hipStreamEndCapture(s->stream, &s->graph);
        if (st == Q3_OK && e == hipSuccess && s->graph) break;
        if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
        (void)hipGetLastError();
        if (attempt >= 3) { Q3C(st); return set_err(Q3_HIP_ERROR, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
    }
    if (!s->aql && !s->aql_tried) {
        s->aql_tried = true;
        const char* e = getenv("Q3_AQL");
        int mode = e ? atoi(e) : 3;
        // Packets without ANY boundary fence give WRONG codes (state that crosses frames moves with plain accesses):
        // mode 2 and the fence halves are probes and need an explicit opt-in, so that a stray environment variable cannot
        // silently corrupt a server's output.
        const bool unsafe_ok = getenv("Q3_AQL_UNSAFE") && atoi(getenv("Q3_AQL_UNSAFE")) == 1;
        const bool wants_unsafe = mode == 2 || getenv("Q3_AQL_ACQ") || getenv("Q3_AQL_REL");
        if (wants_unsafe && !unsafe_ok) {
            static std::atomic<bool> told{false};
            if (!told.exchange(true)) fprintf(stderr, "[q3] Q3_AQL=2 / Q3_AQL_ACQ / Q3_AQL_REL drop kernel-boundary fences and produce wrong results with the product kernels; "
                                                      "ignored without Q3_AQL_UNSAFE=1 (frames stay on %s)\n", mode == 2 ? "hipGraphLaunch" : "HIP's fence policy");
            if (mode == 2) mode = 0;
        }
        if (mode > 0) {
            q3::AqlPolicy pol; pol.fence = mode == 2 ? 0 : 1;
            pol.acquire = pol.release = pol.fence;
            if (mode == 3) pol.node_policy = frame_fence_policy;      // fence-free boundaries between the kernels that move their data write-through
            if (unsafe_ok) {
                if (const char* a = getenv("Q3_AQL_ACQ")) pol.acquire = atoi(a);        // probes: the two fences of a boundary priced separately
                if (const char* r = getenv("Q3_AQL_REL")) pol.release = atoi(r);
                fprintf(stderr, "[q3] WARNING: Q3_AQL_UNSAFE=1: frames are submitted with acquire=%d release=%d — results are NOT valid with the product kernels\n", pol.acquire, pol.release);
            }
            std::string why;
            s->aql = q3::aql_program_create(s->graph, s->m->device, pol, &why);
            if (s->aql) {
                s->aql_mode = mode == 2 ? 2 : mode == 3 ? 3 : 1;
                if (const char* c = getenv("Q3_FRAME_CUS")) { std::string w; if (atoi(c) > 0) q3::aql_restrict_cus(s->aql, atoi(c), &w); }   // measurement aid: the queue keeps this mask
            } else if (getenv("Q3_AQL_VERBOSE")) fprintf(stderr, "[q3] AQL submission unavailable, staying on hipGraphLaunch: %s\n", why.c_str());
        }
    }
    if (!s->aql && !s->graph_exec) HIPC(hipGraphInstantiate(&s->graph_exec, s->graph, nullptr, nullptr, 0));
    return Q3_OK;
}
// `n` replays of the captured frame enqueued WITHOUT waiting for them (streaming read-ahead): on the own queue behind whatever it
// holds, or as graph launches on the session's stream. The caller reserved the K/V pages (kv_reserve_frames) and, on the own queue,
// made sure everything the first frame reads has landed (the queue is not ordered with the HIP stream). sync_frames waits for both.
static q3_status frames_enqueue(q3_session* s, int n) {
    if (n <= 0) return Q3_OK;
    if (s->aql) {
        std::string why; int handed = 0;
        const bool ok = q3::aql_submit(s->aql, n, &why, &handed);
        s->frames_run += handed;
        if (!ok) { s->aql_failed = true; s->codes_host_valid = false; return set_err(Q3_HIP_ERROR, "AQL frame submission: %s", why.c_str()); }
    } else {
        if (!s->graph_exec) return set_err(Q3_INVALID_ARG, "frames_enqueue: no captured frame");
        for (int i = 0; i < n; ++i) HIPC(hipGraphLaunch(s->graph_exec, s->stream));
        s->frames_run += n;
    }
    s->codes_host_valid = false;
    return Q3_OK;
}

extern "C" q3_status q3_session_generate(q3_session* s, int n_frames, int use_graph) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (!s->prefilled) return set_err(Q3_INVALID_ARG, "session not prefilled");
    if (s->aql_failed) return set_err(Q3_HIP_ERROR, "session unusable: an earlier frame submission on the AQL queue failed or timed out");
    HIPC(hipSetDevice(s->m->device));
    if (s->debug || s->profile) use_graph = 0;
    int todo = n_frames;
    { const int left = session_remaining(s); if (todo > left) todo = left; }
    if (todo <= 0) return Q3_OK;
    Q3C(kv_reserve_frames(s, todo));        // paged KV: every page these frames can reach, before the first of them is queued
    if (use_graph) Q3C(frame_capture(s, false));
    bool on_aql = use_graph && s->aql;
    if (on_aql) HIPC(sync_frames(s));     // the queue is not ordered with the HIP stream: prefill / swaps must have landed
    bool eos_on = false;
    for (const auto& q : s->seq) eos_on = eos_on || q.req.opts.eos_token_id >= 0;
    const int check_every = 32;
    while (todo > 0) {
        const int burst = eos_on ? (todo < check_every ? todo : check_every) : todo;
        if (on_aql) {
            // (a failed submission or wait marks the session failed: the ring may still hold — or run — packets that write this
            // session's buffers, so nothing may replay behind them; frames_run stays in step with what the device was given)
            q3_status st = frames_enqueue(s, burst);
            if (st == Q3_OK && sync_frames(s) != hipSuccess) st = Q3_HIP_ERROR;
            if (st != Q3_OK) { s->aql_failed = true; s->codes_host_valid = false; return st; }
        } else
        for (int i = 0; i < burst; ++i) {
            if (use_graph) HIPC(hipGraphLaunch(s->graph_exec, s->stream));
            else Q3C(frame_launch(s));
            s->frames_run += 1;
        }
        todo -= burst;
        s->codes_host_valid = false;
        if (eos_on) { Q3C(refresh_codes(s)); if (all_done(s)) break; }
    }
    HIPC(sync_frames(s));
    if (s->profile && !s->prof_events.empty()) {
        for (size_t i = 0; i < s->prof_events.size(); ++i) {
            float ms = 0; hipEventElapsedTime(&ms, s->prof_events[i].first, s->prof_events[i].second);
            s->prof_linear.ms += ms; s->prof_linear.bytes += s->prof_event_bytes[i]; s->prof_linear.launches += 1;
        }
        s->prof_events.clear(); s->prof_event_bytes.clear(); s->prof_pool_next = 0;
    }
    return Q3_OK;
}

static q3_status decode_range_on(q3_session* s, int b, int f0, int f1, hipStream_t st, float* pcm_host, size_t cap, size_t* n_samples);

// Streaming with several sequences in one session: the next chunk of row b (StreamingSession::next_chunk, lib.rs:1650-1759,
// one per row). Rows advance in lockstep, so asking row after row costs the frames once: the first call generates them for
// every row, the others find theirs buffered and only run their vocoder. One row: q3_session_next_chunk (with read-ahead).
extern "C" q3_status q3_session_next_chunk_row(q3_session* s, int b, float* pcm_host, size_t cap, size_t* n_samples, int* done) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    if (!s->prefilled) Q3C(q3_session_prefill(s));
    Q3C(refresh_codes(s));
    SeqInfo& q = s->seq[b];
    const int chunk = s->opts.chunk_frames > 0 ? s->opts.chunk_frames : 10;
    while (!q.done && q.n_frames - q.stream_pos < chunk && session_remaining(s) > 0) {
        Q3C(q3_session_generate(s, chunk - (q.n_frames - q.stream_pos), 1));
        Q3C(refresh_codes(s));
    }
    int avail = q.n_frames - q.stream_pos;
    if (avail > chunk) avail = chunk;
    if (avail <= 0) { if (n_samples) *n_samples = 0; if (done) *done = 1; return Q3_OK; }
    const int spf = samples_per_frame(s->m->cfg);
    if (s->stream_mode == 1 && !q.icl) {
        // continuous mode (q3_session_set_stream_mode): left context re-run, sample-exact with the whole-utterance decode
        const int a0 = q.stream_pos, e = q.stream_pos + avail, c0 = a0 > CODEC_CTX_FRAMES ? a0 - CODEC_CTX_FRAMES : 0;
        Q3C(codec_reserve(s->m, s->cws, chunk + CODEC_CTX_FRAMES, s->max_frames));
        HIPC(hipMemcpyAsync(s->cws.frames, s->codes + (size_t)b * s->max_frames * 16, (size_t)e * 16 * 4, hipMemcpyDeviceToDevice, s->stream));
        Q3C(codec_decode_dev(s->m, s->cws, e, s->stream, nullptr, c0));
        HIPC(sync_frames(s));
        if (n_samples) *n_samples = (size_t)avail * spf;
        if (pcm_host) {
            if (cap < (size_t)avail * spf) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
            HIPC(q3_hipMemcpy(pcm_host, s->cws.pcm + (size_t)(a0 - c0) * spf, (size_t)avail * spf * 4, hipMemcpyDeviceToHost));
        }
    } else {
        Q3C(decode_range_on(s, b, q.stream_pos, q.stream_pos + avail, s->stream, pcm_host, cap, n_samples));
    }
    q.stream_pos += avail;
    if (done) *done = (q.done && q.stream_pos >= q.n_frames) ? 1 : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_frames(q3_session* s, int b, int* n_frames, int* done) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    Q3C(refresh_codes(s));
    if (n_frames) *n_frames = s->seq[b].n_frames;
    if (done) *done = s->seq[b].done ? 1 : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_codes(q3_session* s, int b, uint32_t* codes_host, int cap_frames, int* n_frames) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    Q3C(refresh_codes(s));
    const int n = s->seq[b].n_frames;
    if (n_frames) *n_frames = n;
    if (codes_host) {
        if (cap_frames < n) return set_err(Q3_INVALID_ARG, "codes buffer too small (%d < %d frames)", cap_frames, n);
        memcpy(codes_host, &s->codes_host[(size_t)b * s->max_frames * 16], (size_t)n * 16 * 4);
    }
    return Q3_OK;
}

// context-free decode of frames [f0, f1) of sequence b on `st` (the reference's per-chunk decode, lib.rs:1755-1758)
static q3_status decode_range_on(q3_session* s, int b, int f0, int f1, hipStream_t st, float* pcm_host, size_t cap, size_t* n_samples) {
    const int T = f1 - f0, spf = samples_per_frame(s->m->cfg);
    if (n_samples) *n_samples = (size_t)T * spf;
    if (T == 0) return Q3_OK;
    Q3C(codec_reserve(s->m, s->cws, T));
    HIPC(hipMemcpyAsync(s->cws.frames, s->codes + ((size_t)b * s->max_frames + f0) * 16, (size_t)T * 16 * 4, hipMemcpyDeviceToDevice, st));
    Q3C(codec_decode_dev(s->m, s->cws, T, st, nullptr));
    HIPC(hipStreamSynchronize(st));
    if (pcm_host) {
        if (cap < (size_t)T * spf) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
        HIPC(q3_hipMemcpy(pcm_host, s->cws.pcm, (size_t)T * spf * 4, hipMemcpyDeviceToHost));
    }
    return Q3_OK;
}

extern "C" q3_status q3_session_decode(q3_session* s, int b, int f0, int f1, float* pcm_host, size_t cap, size_t* n_samples) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    Q3C(refresh_codes(s));
    if (f0 < 0 || f1 < f0 || f1 > s->seq[b].n_frames) return set_err(Q3_INVALID_ARG, "bad frame range [%d,%d) of %d", f0, f1, s->seq[b].n_frames);
    const int T = f1 - f0, spf = samples_per_frame(s->m->cfg);
    const SeqInfo& q = s->seq[b];
    if (!q.ref_codes.empty() && s->prefilled && f0 == 0 && f1 == q.n_frames) {
        // ICL full-utterance decode (lib.rs:1022-1041): decode [ref_frames ; generated], then cut the first
        // ref_len * samples / total_frames samples
        const int n_ref = (int)(q.ref_codes.size() / 16), total = n_ref + T;
        const size_t all = (size_t)total * spf, cut = (size_t)n_ref * all / (size_t)(total > 0 ? total : 1);
        if (n_samples) *n_samples = all - cut;
        Q3C(codec_reserve(s->m, s->cws, total));
        // the row's reference frames from the host copy (a row swapped in by q3_session_replace brings its own)
        HIPC(hipMemcpyAsync(s->cws.frames, q.ref_codes.data(), (size_t)n_ref * 16 * 4, hipMemcpyHostToDevice, s->stream));
        if (T > 0) HIPC(hipMemcpyAsync(s->cws.frames + (size_t)n_ref * 16, s->codes + (size_t)b * s->max_frames * 16, (size_t)T * 16 * 4, hipMemcpyDeviceToDevice, s->stream));
        Q3C(codec_decode_dev(s->m, s->cws, total, s->stream, nullptr));
        HIPC(sync_frames(s));
        if (pcm_host) {
            if (cap < all - cut) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
            HIPC(q3_hipMemcpy(pcm_host, s->cws.pcm + cut, (all - cut) * 4, hipMemcpyDeviceToHost));
        }
        return Q3_OK;
    }
    return decode_range_on(s, b, f0, f1, s->stream, pcm_host, cap, n_samples);
}

// synthesize_with_timing for the whole batch. With Q3_DECODE_OVERLAP=1 (and no ICL sequence) the vocoder does not
// wait for the last frame: every Q3_DECODE_SEG (default 128) generated frames a helper thread enqueues the segment's
// decode (exact: CODEC_CTX_FRAMES of left context re-run, see codec_decode_dev) on a second stream beside the frame
// loop, the samples going straight to the caller's buffers; the last Q3_DECODE_TAIL (32) frames of a row are the only
// segment nothing overlaps. OFF by default — measured again in round 6 (1.7B, 8 x 640 frames, frames on the own queue):
// the frame loop slows from 1636 to 1767 ms while the decode tail shrinks from 155 to 54 ms (2843 -> 2797 frames/s); the
// frame loop's workgroups need a whole CU's registers, so vocoder waves already resident on a CU hold them up, and
// splitting the chip between the two (Q3_DECODE_CUS below) costs the frames more than the vocoder takes.
extern "C" q3_status q3_session_run(q3_session* s, int use_graph, float** pcm_host, const size_t* cap, size_t* n_samples, q3_timing* timing) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    s->precapture = use_graph != 0;
    Q3C(q3_session_prefill(s));
    const auto t1 = clk::now();
    const int overlap_env = [] { const char* e = getenv("Q3_DECODE_OVERLAP"); return e ? atoi(e) : 0; }();     // read per call: opt-in, tests set it per test
    static const int seg_env = [] { const char* e = getenv("Q3_DECODE_SEG"); const int v = e ? atoi(e) : 128; return v < 16 ? 16 : v; }();
    bool overlap = overlap_env != 0 && !s->debug && !s->profile && s->max_frames > seg_env;
    for (auto& q : s->seq) if (!q.ref_codes.empty()) overlap = false;
    const int spf = samples_per_frame(s->m->cfg);
    if (!overlap) {
        Q3C(q3_session_generate(s, s->max_frames, use_graph));
        Q3C(refresh_codes(s));
        const auto t2 = clk::now();
        int total = 0;
        // Two utterances are vocoded at a time, each on its own stream and workspace: most of a decode saturates the
        // chip, but its front (the pre-transformer's ~90 launches on 10-160 workgroups) is latency-bound and fills in beside
        // the other utterance's convolutions. Q3_DECODE_PAIRS=n: n at a time (1 = serial; A/B aid). Re-measured in round 6
        // (profiles/r6_decode_knobs_ab.txt): 8 x 640 frames 155.3 / 157.7 / 162.1 ms at 2 / 4 / 8 at a time, 64 x 640 frames
        // 1261 / 1274 ms at 2 / 4 — four was the round-1 optimum, when the front was 4 of 28 ms; it is 2 of 19.8 now.
        static const int conc = [] { const char* e = getenv("Q3_DECODE_PAIRS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
        bool any_icl = false;
        for (auto& q : s->seq) any_icl = any_icl || !q.ref_codes.empty();
        if (conc > 1 && s->B > 1 && !any_icl) {
            while ((int)s->par_ws.size() < conc - 1) {
                s->par_ws.emplace_back();
                hipStream_t st = nullptr;
                HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                s->par_streams.push_back(st);
            }
            auto ws_of = [&](int k) -> CodecWS& { return k == 0 ? s->cws : s->par_ws[(size_t)k - 1]; };
            auto st_of = [&](int k) { return k == 0 ? s->stream : s->par_streams[(size_t)k - 1]; };
            if (pcm_host)
                for (int b = 0; b < s->B; ++b)
                    if (pcm_host[b] && s->seq[b].n_frames && (!cap || cap[b] < (size_t)s->seq[b].n_frames * spf)) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
            for (int b0 = 0; b0 < s->B; b0 += conc) {
                const int nb = (s->B - b0) < conc ? (s->B - b0) : conc;
                for (int k = 0; k < nb; ++k) {
                    const int b = b0 + k, T = s->seq[b].n_frames;
                    if (n_samples) n_samples[b] = (size_t)T * spf;
                    total += T;
                    if (T == 0) continue;
                    Q3C(codec_reserve(s->m, ws_of(k), T));
                    HIPC(hipMemcpyAsync(ws_of(k).frames, s->codes + (size_t)b * s->max_frames * 16, (size_t)T * 16 * 4, hipMemcpyDeviceToDevice, st_of(k)));
                    Q3C(codec_decode_dev(s->m, ws_of(k), T, st_of(k), nullptr));
                    // the samples leave for the host on the utterance's own stream, beside the other utterances' decodes
                    // (synthesize returns host samples, lib.rs:718-784); pinned caller buffers make this a true async copy
                    if (pcm_host && pcm_host[b])
                        HIPC(hipMemcpyAsync(pcm_host[b], ws_of(k).pcm, (size_t)T * spf * 4, hipMemcpyDeviceToHost, st_of(k)));
                }
                for (int k = 0; k < nb; ++k) HIPC(hipStreamSynchronize(st_of(k)));
            }
        } else
        for (int b = 0; b < s->B; ++b) {
            size_t n = 0;
            Q3C(q3_session_decode(s, b, 0, s->seq[b].n_frames, pcm_host ? pcm_host[b] : nullptr, cap ? cap[b] : 0, &n));
            if (n_samples) n_samples[b] = n;
            total += s->seq[b].n_frames;
        }
        const auto t3 = clk::now();
        if (timing) { timing->prefill_ms = ms(t0, t1); timing->generation_ms = ms(t1, t2); timing->decode_ms = ms(t2, t3); timing->generation_frames = total; }
        return Q3_OK;
    }
    HIPC(hipSetDevice(s->m->device));
    // Q3_DECODE_CUS=n (measurement aid, default 0 = off) SPLITS the chip while both run: the frames keep the first 256 - n bits of the CU
    // mask on their own queue, the decode stream gets the last n (n / 8 CUs of every XCD, see aql_restrict_cus), so a frame kernel's
    // workgroups are never dealt onto a CU the vocoder holds. Measured in round 6 (profiles/r6_split_chip_overlap_ab.txt): the vocoder
    // beside the frames then costs them only 3 %, but the frames ALONE on 224 / 192 / 128 CUs are 16 / 15 / 23 % slower — their grids are
    // one workgroup per CU of the whole chip, and the slowest workgroup of every one of 549 dependent launches sets the pace.
    static const int cus = [] { const char* e = getenv("Q3_DECODE_CUS"); const int v = e ? atoi(e) : 0; return v <= 0 ? 0 : (v < 8 ? 8 : (v > 248 ? 248 : v & ~7)); }();
    if (!s->dec_stream) {
        if (cus) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 256 - cus; i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
            HIPC(hipExtStreamCreateWithCUMask(&s->dec_stream, 8, mask));
        } else {
            int least = 0, greatest = 0;
            HIPC(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPC(hipStreamCreateWithPriority(&s->dec_stream, hipStreamNonBlocking, least));
        }
    }
    Q3C(codec_reserve(s->m, s->seg_ws, seg_env + CODEC_CTX_FRAMES, s->max_frames));
    struct Split {                                         // the whole chip goes back to the frames' queue on every way out
        q3::AqlProgram* p = nullptr;
        void release() { if (p) { std::string why; q3::aql_restrict_cus(p, 0, &why); p = nullptr; } }
        ~Split() { release(); }
    } split;
    if (cus && s->aql && !s->aql_failed) { std::string why; if (q3::aql_restrict_cus(s->aql, 256 - cus, &why)) split.p = s->aql; }
    std::vector<int> dec_pos((size_t)s->B, 0);
    std::thread worker; q3_status wst = Q3_OK; std::string werr;
    auto join = [&]() -> q3_status {
        if (worker.joinable()) worker.join();
        if (wst != Q3_OK) return set_err(wst, "%s", werr.c_str());
        return Q3_OK;
    };
    struct Job { int b, a, e; };
    // one segment: codes -> workspace, vocoder over [a - context, e), the samples of [a, e) straight to the caller's buffer
    auto seg_enqueue = [s, spf, pcm_host, cap](const Job& j, CodecWS& ws, hipStream_t st) -> q3_status {
        const int c0 = j.a > CODEC_CTX_FRAMES ? j.a - CODEC_CTX_FRAMES : 0;
        if (pcm_host && pcm_host[j.b] && (!cap || cap[j.b] < (size_t)j.e * spf)) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
        HIPC(hipMemcpyAsync(ws.frames, s->codes + (size_t)j.b * s->max_frames * 16, (size_t)j.e * 16 * 4, hipMemcpyDeviceToDevice, st));
        Q3C(codec_decode_dev(s->m, ws, j.e, st, nullptr, c0));
        if (pcm_host && pcm_host[j.b])
            HIPC(hipMemcpyAsync(pcm_host[j.b] + (size_t)j.a * spf, ws.pcm + (size_t)(j.a - c0) * spf, (size_t)(j.e - j.a) * spf * 4, hipMemcpyDeviceToHost, st));
        return Q3_OK;
    };
    auto collect = [&](bool final_pass) {
        std::vector<Job> jobs;
        for (int b = 0; b < s->B; ++b) {
            const SeqInfo& q = s->seq[b];
            const int e = (q.done || final_pass) ? q.n_frames : (q.n_frames < s->frames_run ? q.n_frames : s->frames_run);
            if (e > dec_pos[(size_t)b] && (final_pass || q.done || e - dec_pos[(size_t)b] >= 16)) {
                // keep every call within the workspace: at most seg_env new frames per job
                for (int a = dec_pos[(size_t)b]; a < e; a += seg_env) jobs.push_back({b, a, a + seg_env < e ? a + seg_env : e});
                dec_pos[(size_t)b] = e;
            }
        }
        return jobs;
    };
    auto dispatch = [&]() -> q3_status {
        std::vector<Job> jobs = collect(false);
        if (jobs.empty()) return Q3_OK;
        Q3C(join());
        worker = std::thread([s, jobs, seg_enqueue, &wst, &werr]() {
            if (hipSetDevice(s->m->device) != hipSuccess) { wst = Q3_HIP_ERROR; werr = "hipSetDevice failed in the decode thread"; return; }
            for (const Job& j : jobs) {
                const q3_status st = seg_enqueue(j, s->seg_ws, s->dec_stream);
                if (st != Q3_OK) { wst = st; werr = q3_last_error(); return; }
            }
        });
        return Q3_OK;
    };
    // the last segment of a row is the only one nothing overlaps: it is kept short (Q3_DECODE_TAIL frames, default 32)
    static const int tail_env = [] { const char* e = getenv("Q3_DECODE_TAIL"); const int v = e ? atoi(e) : 32; return v < 16 ? 16 : v; }();
    q3_status st = Q3_OK;
    while (st == Q3_OK && session_remaining(s) > 0 && !all_done(s)) {
        int n = seg_env;
        const int left = session_remaining(s);
        if (left > tail_env && left - n < tail_env) n = left - tail_env;        // ... so the step before it stops tail_env frames short of the end
        st = q3_session_generate(s, n, use_graph);
        if (st == Q3_OK) st = refresh_codes(s);
        if (st == Q3_OK && session_remaining(s) > 0 && !all_done(s)) st = dispatch();
    }
    const auto t2 = clk::now();
    split.release();
    { const q3_status js = join(); if (st == Q3_OK) st = js; }
    if (st != Q3_OK) { hipStreamSynchronize(s->dec_stream); return st; }
    // what is left runs on the whole chip beside the confined stream's backlog: utterances side by side as in the plain path.
    // Whatever happens, nothing returns while a stream may still be copying samples into the caller's buffers.
    {
        const int conc = 2;
        auto ws_of = [&](int k) -> CodecWS& { return k == 0 ? s->cws : s->par_ws[(size_t)k - 1]; };
        auto st_of = [&](int k) { return k == 0 ? s->stream : s->par_streams[(size_t)k - 1]; };
        auto tail = [&]() -> q3_status {
            std::vector<Job> jobs = collect(true);
            while ((int)s->par_ws.size() < conc - 1) {
                hipStream_t pst = nullptr;
                HIPC(hipStreamCreateWithFlags(&pst, hipStreamNonBlocking));
                s->par_ws.emplace_back(); s->par_streams.push_back(pst);
            }
            for (int k = 0; k < conc; ++k) Q3C(codec_reserve(s->m, ws_of(k), seg_env + CODEC_CTX_FRAMES, s->max_frames));
            for (size_t i = 0; i < jobs.size(); ++i) Q3C(seg_enqueue(jobs[i], ws_of((int)(i % conc)), st_of((int)(i % conc))));
            return Q3_OK;
        };
        st = tail();
        hipError_t he = hipStreamSynchronize(s->dec_stream);
        for (int k = 0; k < conc && k <= (int)s->par_streams.size(); ++k) { const hipError_t e2 = hipStreamSynchronize(st_of(k)); if (he == hipSuccess) he = e2; }
        if (st != Q3_OK) return st;
        HIPC(he);
    }
    int total = 0;
    for (int b = 0; b < s->B; ++b) {
        if (n_samples) n_samples[b] = (size_t)s->seq[b].n_frames * spf;
        total += s->seq[b].n_frames;
    }
    const auto t3 = clk::now();
    if (timing) { timing->prefill_ms = ms(t0, t1); timing->generation_ms = ms(t1, t2); timing->decode_ms = ms(t2, t3); timing->generation_frames = total; }
    return Q3_OK;
}

extern "C" q3_status q3_session_set_stream_mode(q3_session* s, int mode) {
    if (!s || (mode != 0 && mode != 1)) return set_err(Q3_INVALID_ARG, "q3_session_set_stream_mode: mode must be 0 (context-free) or 1 (continuous)");
    s->stream_mode = mode;
    return Q3_OK;
}

extern "C" q3_status q3_session_next_chunk(q3_session* s, float* pcm_host, size_t cap, size_t* n_samples, int* done) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->B != 1) return set_err(Q3_UNSUPPORTED, "streaming sessions are batch 1 (StreamingSession, lib.rs:1484)");
    if (!s->prefilled) { s->precapture = true; Q3C(q3_session_prefill(s)); }
    Q3C(refresh_codes(s));
    SeqInfo& q = s->seq[0];
    const int chunk = s->opts.chunk_frames > 0 ? s->opts.chunk_frames : 10;
    // generate until chunk_frames frames are buffered or the sequence ends (lib.rs:1663-1748)
    while (!q.done && q.n_frames - s->stream_pos < chunk && session_remaining(s) > 0) {
        int need = chunk - (q.n_frames - s->stream_pos);
        Q3C(q3_session_generate(s, need, 1));
        Q3C(refresh_codes(s));
    }
    int avail = q.n_frames - s->stream_pos;
    if (avail > chunk) avail = chunk;
    if (avail <= 0) { if (n_samples) *n_samples = 0; if (done) *done = 1; return Q3_OK; }
    const bool chunk_done = q.done && s->stream_pos + avail >= q.n_frames;
    // Read-ahead: the frames of the NEXT chunk are enqueued (graph replays, no host wait) while this chunk is vocoded
    // and handed over, so the frame loop keeps going while the host copies out, returns and plays the chunk. First
    // chunk: its vocoder is enqueued first and the replays behind it on the same stream (time-to-first-audio is what it
    // was; the host waits on an event, not on the stream). Later chunks: the replays go first and the chunk's vocoder
    // runs beside them on its own stream — a chunk then costs max(generation, decode) instead of their sum (streaming
    // RTF 0.046 -> 0.042 on the 1.7B model). After EOS the device-side done flag turns extra replays into no-ops.
    // Q3_STREAM_NO_AHEAD=1 restores the serial schedule (A/B aid).
    static const bool no_ahead = getenv("Q3_STREAM_NO_AHEAD") != nullptr;
    int ahead = 0;
    if (!no_ahead && !q.done && (s->aql || s->graph_exec) && !s->aql_failed && !s->debug && !s->profile) {
        ahead = chunk - (q.n_frames - (s->stream_pos + avail));
        if (ahead > session_remaining(s)) ahead = session_remaining(s);
        if (ahead < 0) ahead = 0;
    }
    const bool first = s->stream_pos == 0;
    auto launch_ahead = [&]() -> q3_status {
        Q3C(kv_reserve_frames(s, ahead));                        // (its table updates ride s->stream)
        if (s->aql) HIPC(hipStreamSynchronize(s->stream));       // the own queue is not ordered with the stream
        return frames_enqueue(s, ahead);                         // the next call re-reads codes / EOS state after sync_frames
    };
    hipStream_t dst = s->stream;
    if (ahead > 0 && !first) {
        if (!s->dec_stream) HIPC(hipStreamCreateWithFlags(&s->dec_stream, hipStreamNonBlocking));
        Q3C(launch_ahead());
        dst = s->dec_stream;
    }
    // enqueue this chunk's vocoder on dst
    const int spf = samples_per_frame(s->m->cfg);
    const float* src = nullptr;
    if (s->stream_mode == 1 && !q.icl) {
        // continuous mode: the front runs over frames [0, end), the convolutional stack over [pos - CTX, end); the chunk's
        // samples are identical to the same frames of a whole-utterance decode (codec_decode_dev)
        const int a0 = s->stream_pos, e = s->stream_pos + avail;
        const int c0 = a0 > CODEC_CTX_FRAMES ? a0 - CODEC_CTX_FRAMES : 0;
        Q3C(codec_reserve(s->m, s->cws, chunk + CODEC_CTX_FRAMES, s->max_frames));
        HIPC(hipMemcpyAsync(s->cws.frames, s->codes, (size_t)e * 16 * 4, hipMemcpyDeviceToDevice, dst));
        Q3C(codec_decode_dev(s->m, s->cws, e, dst, nullptr, c0));
        src = s->cws.pcm + (size_t)(a0 - c0) * spf;
    } else {
        // the reference's schedule: the chunk decoded as an independent utterance (lib.rs:1755-1758)
        Q3C(codec_reserve(s->m, s->cws, avail));
        HIPC(hipMemcpyAsync(s->cws.frames, s->codes + (size_t)s->stream_pos * 16, (size_t)avail * 16 * 4, hipMemcpyDeviceToDevice, dst));
        Q3C(codec_decode_dev(s->m, s->cws, avail, dst, nullptr));
        src = s->cws.pcm;
    }
    if (ahead > 0 && first) {
        if (!s->dec_ev) HIPC(hipEventCreateWithFlags(&s->dec_ev, hipEventDisableTiming));
        HIPC(hipEventRecord(s->dec_ev, s->stream));
        Q3C(launch_ahead());
        HIPC(hipEventSynchronize(s->dec_ev));
    } else {
        HIPC(hipStreamSynchronize(dst));
    }
    if (n_samples) *n_samples = (size_t)avail * spf;
    if (pcm_host) {
        if (cap < (size_t)avail * spf) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
        HIPC(q3_hipMemcpy(pcm_host, src, (size_t)avail * spf * 4, hipMemcpyDeviceToHost));
    }
    s->stream_pos += avail;
    if (done) *done = chunk_done ? 1 : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_get(q3_session* s, int what, int b, void* out, size_t bytes) {
    if (!s || !out || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad argument");
    const q3_config& c = s->m->cfg;
    HIPC(hipSetDevice(s->m->device));
    HIPC(sync_frames(s));
    const void* src = nullptr; size_t need = 0;
    switch (what) {
        case Q3_GET_PREFILL_EMBEDS: src = s->embeds + (size_t)b * s->prefill_len * c.hidden; need = (size_t)s->prefill_len * c.hidden * 4; break;
        case Q3_GET_LAST_HIDDEN: src = s->LASTH + (size_t)b * c.hidden; need = (size_t)c.hidden * 4; break;
        case Q3_GET_LOGITS: src = s->LOGITS + (size_t)b * c.codec_vocab; need = (size_t)c.codec_vocab * 4; break;
        case Q3_GET_TRAILING: src = s->rows + (size_t)s->seq[b].trail_base * c.hidden; need = (size_t)s->seq[b].trailing_len * c.hidden * 4; break;
        case Q3_GET_PAD_EMBED: src = s->rows + (size_t)s->seq[b].pad_row * c.hidden; need = (size_t)c.hidden * 4; break;
        case Q3_GET_LOGITS_HIST:
            if (!s->logits_hist) return set_err(Q3_INVALID_ARG, "session has no debug capture");
            src = s->logits_hist + (size_t)b * (s->max_frames + 1) * c.codec_vocab; need = (size_t)(s->frames_run + 1) * c.codec_vocab * 4; break;
        case Q3_GET_TOKEN: src = s->tok + b; need = 4; break;
        case Q3_GET_CP_LOGITS: {
            // [15][B][V] on device → [15][V] for sequence b
            need = (size_t)15 * c.cp_vocab * 4;
            if (bytes < need) return set_err(Q3_INVALID_ARG, "buffer too small");
            for (int g = 0; g < 15; ++g)
                HIPC(q3_hipMemcpy((char*)out + (size_t)g * c.cp_vocab * 4, s->CP_LOGITS + ((size_t)g * s->B + b) * c.cp_vocab, (size_t)c.cp_vocab * 4, hipMemcpyDeviceToHost));
            return Q3_OK;
        }
        case Q3_GET_CP_LOGITS_HIST: {
            if (!s->cp_logits_hist) return set_err(Q3_INVALID_ARG, "session has no debug capture");
            need = (size_t)s->frames_run * 15 * c.cp_vocab * 4;
            if (bytes < need) return set_err(Q3_INVALID_ARG, "buffer too small");
            for (int f = 0; f < s->frames_run; ++f)
                for (int g = 0; g < 15; ++g)
                    HIPC(q3_hipMemcpy((char*)out + ((size_t)f * 15 + g) * c.cp_vocab * 4,
                                   s->cp_logits_hist + (((size_t)f * 15 + g) * s->B + b) * c.cp_vocab, (size_t)c.cp_vocab * 4, hipMemcpyDeviceToHost));
            return Q3_OK;
        }
        default: return set_err(Q3_INVALID_ARG, "unknown item %d", what);
    }
    if (bytes < need) return set_err(Q3_INVALID_ARG, "buffer too small (%zu < %zu)", bytes, need);
    HIPC(q3_hipMemcpy(out, src, need, hipMemcpyDeviceToHost));
    return Q3_OK;
}

