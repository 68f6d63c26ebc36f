// q3_batcher.hip — continuous batching: q3_session_replace (side prefill + transplant) and the native batcher q3_batcher_*
// (one of the five units of the engine: q3_engine.h says which holds what)
#include "q3_engine.h"
#include <condition_variable>
#include <deque>

// Continuous batching: swap a finished row of a running session for a new request (include/q3tts.h). The reference keeps all
// per-utterance state per call (KV caches, sampling context, penalty mask, trailing text: lib.rs:743-756, 1484-1541); here
// that state is the row's slice of the session's device arrays, so a swap = prefill the request in a one-row side session
// (the unchanged prefill path) and copy its slice in: K/V extents of the prompt positions, last hidden state, first sampled
// token, penalty mask, counters, the pre-drawn PCG stream, projected text rows. The captured frame graph is untouched — it
// only ever reads these arrays — and the other rows do not notice: their state, and therefore their bits, are unchanged.
// Row j of a prefilled side session becomes row b of the host session: the per-row state the captured frame graph reads is
// copied in, the prompt's K/V pages are relinked (contiguous extents: copied). Both streams are idle (the caller drained them).
q3_status transplant_row(q3_session* s, int b, q3_session* side, int j, int limit) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const SeqInfo& sq = side->seq[(size_t)j];
    const int H = c.hidden, S = sq.prefill_len, nkv = c.n_kv_heads;
    const size_t row_bytes = (size_t)HEAD_DIM * 4;
    if (s->paged) {
        // the prompt's K/V is not copied: the side session's pages become the row's (its old ones go back to the pool), and the
        // row's table entries are rewritten
        kv_release_row(s, b);
        std::vector<float*>& row = s->kv_rows[(size_t)b];
        row.assign(side->kv_rows[(size_t)j].begin(), side->kv_rows[(size_t)j].end());
        side->kv_rows[(size_t)j].clear();
        HIPC(hipMemcpyAsync(s->kv_table + (size_t)b * KV_MAX_PAGES, row.data(), row.size() * 8, hipMemcpyHostToDevice, s->stream));
    } else
    for (int l = 0; l < c.n_layers; ++l) {
        const size_t so = (size_t)l * side->kv_layer_stride + (size_t)j * nkv * side->max_seq * HEAD_DIM, dof = (size_t)l * s->kv_layer_stride + (size_t)b * nkv * s->max_seq * HEAD_DIM;
        HIPC(hipMemcpy2DAsync(s->kcache + dof, s->max_seq * row_bytes, side->kcache + so, side->max_seq * row_bytes, S * row_bytes, nkv, hipMemcpyDeviceToDevice, s->stream));
        HIPC(hipMemcpy2DAsync(s->vcache + dof, s->max_seq * row_bytes, side->vcache + so, side->max_seq * row_bytes, S * row_bytes, nkv, hipMemcpyDeviceToDevice, s->stream));
    }
    auto d2d = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s->stream); };
    HIPC(d2d(s->LASTH + (size_t)b * H, side->LASTH + (size_t)j * H, (size_t)H * 4));
    HIPC(d2d(s->tok + b, side->tok + j, 4));
    HIPC(d2d(s->seen + (size_t)b * c.codec_vocab, side->seen + (size_t)j * c.codec_vocab, (size_t)c.codec_vocab));
    HIPC(d2d(s->token_count + b, side->token_count + j, 4));
    HIPC(d2d(s->pos + b, side->pos + j, 4));
    HIPC(d2d(s->frame_idx + b, side->frame_idx + j, 4));
    // the row's pre-drawn PCG stream: the side session drew max_frames(side) + 1 >= limit + 1 of them (an ICL cap may make it the shorter one)
    HIPC(d2d(s->U + (size_t)b * (s->max_frames + 2), side->U + (size_t)j * (side->max_frames + 2),
             (size_t)((side->max_frames < s->max_frames ? side->max_frames : s->max_frames) + 2) * 4));
    const int row0 = s->repl_base + b * s->row_cap;
    HIPC(d2d(s->rows + (size_t)row0 * H, side->rows + (size_t)sq.row_base * H, (size_t)sq.n_rows * H * 4));
    const int hv[4] = {row0 + (sq.trail_base - sq.row_base), sq.trailing_len, row0 + (sq.pad_row - sq.row_base), limit};
    HIPC(hipMemcpyAsync(s->trail_base + b, &hv[0], 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->trail_len + b, &hv[1], 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->pad_row + b, &hv[2], 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->limit + b, &hv[3], 4, hipMemcpyHostToDevice, s->stream));
    const SampleRow srow = sample_row(sq.req.opts);
    HIPC(hipMemcpyAsync(s->sample_rows + b, &srow, sizeof srow, hipMemcpyHostToDevice, s->stream));
    HIPC(sync_frames(s));
    SeqInfo nq = sq;
    nq.row_base = row0; nq.trail_base = hv[0]; nq.pad_row = hv[2];
    nq.start_run = s->frames_run; nq.limit = limit; nq.n_frames = 0; nq.done = false; nq.stream_pos = 0; nq.req.opts.max_length = limit; nq.idle = false;
    s->seq[(size_t)b] = nq;
    {   // the request's arrays live in the row's own vectors (the caller's pointers need not outlive the call)
        SeqInfo& q = s->seq[(size_t)b];
        q.req.text_ids = q.text.data(); q.req.instruct_ids = q.instruct.data(); q.req.ref_codes = q.ref_codes.data();
        q.req.ref_text_ids = q.ref_text.data(); q.req.xvector = q.xvec.empty() ? nullptr : q.xvec.data();
    }
    if (b == 0) s->stream_pos = 0;       // q3_session_next_chunk (the row-0 streaming call) starts over with the new utterance too
    s->codes_host_valid = false;
    return Q3_OK;
}
// what a side session must satisfy before its rows may enter the host session
q3_status transplant_check(q3_session* s, q3_session* side, int j, int limit_req, int* limit_out) {
    const SeqInfo& sq = side->seq[(size_t)j];
    if (side->opts.chunk_frames != s->opts.chunk_frames) return set_err(Q3_UNSUPPORTED, "q3_session_replace: chunk_frames is a property of the session");
    const int limit = limit_req < sq.limit ? limit_req : sq.limit;
    if (sq.n_rows > s->row_cap) return set_err(Q3_UNSUPPORTED, "q3_session_replace: the request's %d text rows exceed the session's slot (%d rows: 1024, or the longest text of the original batch)", sq.n_rows, s->row_cap);
    if (s->paged != side->paged) return set_err(Q3_UNSUPPORTED, "q3_session_replace: the sessions disagree on KV paging");
    if (s->kv_bf16 != s->kv_in_bf16) return set_err(Q3_UNSUPPORTED, "q3_session_replace: the session's K/V conversion has not happened yet");
    // max_seq bounds a row in both layouts: the captured frame was specialised for it (key splits, the page-table form of the
    // attention kernel); with pages it reserves nothing — only the pages a row really reaches are taken from the pool
    if (sq.prefill_len + limit + 1 > s->max_seq) return set_err(Q3_KV_OVERFLOW, "q3_session_replace: prompt of %d positions + %d frames exceeds the row's KV extent (%d)", sq.prefill_len, limit, s->max_seq);
    *limit_out = limit;
    return Q3_OK;
}

extern "C" q3_status q3_session_replace(q3_session* s, int b, const q3_request* req) {
    if (!s || !req || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "q3_session_replace: bad argument");
    if (!s->prefilled) return set_err(Q3_INVALID_ARG, "q3_session_replace: session not prefilled");
    if (s->debug || s->profile) return set_err(Q3_UNSUPPORTED, "q3_session_replace: not on debug / profiling sessions");
    const q3_model* m = s->m;
    HIPC(hipSetDevice(m->device));
    q3_request r = *req;
    const int limit_req = r.opts.max_length;
    if (limit_req < 1 || limit_req > s->max_frames) return set_err(Q3_UNSUPPORTED, "q3_session_replace: max_length %d outside 1..%d (the session's frame budget)", limit_req, s->max_frames);
    r.opts.max_length = s->max_frames;                 // the side session draws the row's PCG stream with the host session's stride
    static const bool timing = getenv("Q3_REPLACE_TIMING") != nullptr;      // development aid: where a swap's milliseconds go
    const auto tp0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[q3 replace] %-8s %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
    };
    q3_session* side_raw = nullptr;
    Q3C(session_create(s->m, &r, 1, 0, 0, &side_raw, s->stream));      // on the host's stream: its frames and this prefill are serial anyway
    std::unique_ptr<q3_session> side(side_raw);
    side->kv_bf16 = s->kv_bf16;                        // the side session prefills in f32 and converts, as the host session did
    lap("create");
    // (sampling options are per row — SampleRow —, resolved by the side session: an ICL request's repetition-penalty floor and
    // length cap, lib.rs:913-929, come along)
    int limit = 0;
    Q3C(transplant_check(s, side.get(), 0, limit_req, &limit));
    Q3C(q3_session_prefill(side.get()));               // ends with a synchronisation of the side stream
    lap("prefill");
    HIPC(sync_frames(s));             // no frame of the host session in flight while its row changes
    Q3C(transplant_row(s, b, side.get(), 0, limit));
    lap("copies");
    side.reset();
    lap("free");
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// Continuous batcher: a queue of requests through the rows of ONE session (the native form of what a serving loop does
// with q3_session_replace). No thread of its own: the host calls q3_batcher_step from its loop — submit / step / poll /
// fetch may interleave freely (one thread at a time). A step fills free rows from the queue (one-row side prefill + state
// copy, rows of any prompt kind), runs up to n_frames frames of the shared frame graph, and collects the rows that ended
// (codes, and the PCM if the request asked for it). Every request gets the bits of its own batch-1 run.
// ------------------------------------------------------------------------------------------------
struct BatTicket {
    BatReq req; int state = Q3_TICKET_QUEUED; int row = -1; bool want_pcm = false;
    std::vector<uint32_t> codes; std::vector<float> pcm; int n_frames = 0;
    q3_status st = Q3_OK; std::string err;
    // vocoded by the batcher's decode worker (below): queued / running there until dec_done; the row it ran in is long refilled
    bool decoding = false; std::atomic<bool> dec_done{false}; q3_status dec_st = Q3_OK; std::string dec_err;
};
struct q3_batcher {
    q3_model* m = nullptr; int slots = 0, frame_budget = 0, prompt_budget = 0, chunk_frames = 0;
    q3_session* s = nullptr;
    std::vector<int64_t> owner;                       // ticket running in each row, -1 = free
    std::vector<long> commit;                         // KvBudget units row r may still come to hold (worst case of its request), 0 = free row
    std::vector<int64_t> queue;                       // FIFO of waiting tickets
    std::unordered_map<int64_t, std::unique_ptr<BatTicket>> t;
    int64_t next_id = 1;
    // Round 6: the head of the queue is prefilled AHEAD of the row it will enter. A worker thread opens its one-row side session
    // and runs the (unchanged) prefill on a stream of its own while the captured frame keeps replaying for the live rows — the
    // frame leaves most of the chip idle —; when a row ends, the swap is only the state copy (transplant_row) at that frame
    // boundary. The other rows used to stand still for the whole side prefill (1.9 ms for a short prompt, 45 ms for a 4k-token
    // one). Bits are unchanged: the same kernels on the same inputs, only on another stream. Off under a page limit
    // (q3_model_kv_pool_limit: admission must see a row's pages when it decides) and with Q3_BAT_NO_STAGE=1 (A/B aid).
    struct Stage {
        int64_t id = -1; std::thread thr; q3_session* side = nullptr; q3_status st = Q3_OK; std::string err; int limit = 0;
    } stage;
    std::unique_ptr<struct BatDecoder> dec;           // the decode worker of finished rows (below)
};
// Round 6: a finished row's vocoder no longer stalls the session either. bat_collect used to decode the row's samples on the session's
// own stream before the row could be refilled — every live row stood still for ~20 ms per 640 frames. The codes are on the host
// by then, so the decode goes to a worker thread with its own stream and workspace (one decode at a time, in order), the row is
// idled and refilled at once, and the ticket counts as RUNNING until its samples have landed (poll; fetch waits for them). Same
// kernels on the same codes as q3_session_decode: the same samples. ICL rows (reference frames prepended and cut, lib.rs:1022-1041)
// and Q3_BAT_SYNC_DECODE=1 keep the synchronous decode.
struct BatDecoder {
    std::thread thr; std::mutex mu; std::condition_variable cv, cv_done; std::deque<BatTicket*> q; bool stop = false;
    hipStream_t stream = nullptr; CodecWS ws;
};
static void decoder_main(q3_batcher* b);
static void decoder_push(q3_batcher* b, BatTicket* t) {
    BatDecoder& d = *b->dec;
    std::lock_guard<std::mutex> g(d.mu);
    if (!d.thr.joinable()) d.thr = std::thread(decoder_main, b);
    d.q.push_back(t);
    d.cv.notify_one();
}
static void ticket_wait_decode(q3_batcher* b, BatTicket& t) {
    if (!t.decoding) return;
    BatDecoder& d = *b->dec;
    {
        std::unique_lock<std::mutex> lk(d.mu);
        d.cv_done.wait(lk, [&] { return t.dec_done.load(); });
    }
    t.decoding = false;
    if (t.dec_st != Q3_OK) { t.state = Q3_TICKET_FAILED; t.st = t.dec_st; t.err = t.dec_err; t.pcm.clear(); }
    else t.state = Q3_TICKET_DONE;
}
static void decoder_stop(q3_batcher* b) {
    BatDecoder& d = *b->dec;
    {
        std::lock_guard<std::mutex> g(d.mu);
        d.stop = true; d.cv.notify_all();
    }
    if (d.thr.joinable()) d.thr.join();
    if (d.stream) { (void)hipStreamSynchronize(d.stream); (void)hipStreamDestroy(d.stream); d.stream = nullptr; }
    d.ws.release();
}
static q3_status decoder_run(q3_batcher* b, BatTicket& t) {
    BatDecoder& d = *b->dec;
    const q3_model* m = b->m;
    HIPC(hipSetDevice(m->device));
    if (!d.stream) HIPC(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    const int n = t.n_frames;
    Q3C(codec_reserve(m, d.ws, n));
    HIPC(hipMemcpyAsync(d.ws.frames, t.codes.data(), (size_t)n * 16 * 4, hipMemcpyHostToDevice, d.stream));
    Q3C(codec_decode_dev(m, d.ws, n, d.stream, nullptr));
    HIPC(hipMemcpyAsync(t.pcm.data(), d.ws.pcm, t.pcm.size() * 4, hipMemcpyDeviceToHost, d.stream));
    HIPC(hipStreamSynchronize(d.stream));
    return Q3_OK;
}
static void decoder_main(q3_batcher* b) {
    BatDecoder& d = *b->dec;
    for (;;) {
        BatTicket* t = nullptr;
        {
            std::unique_lock<std::mutex> lk(d.mu);
            d.cv.wait(lk, [&] { return d.stop || !d.q.empty(); });
            if (d.q.empty()) return;                 // stop, and nothing left to decode
            t = d.q.front(); d.q.pop_front();
        }
        const q3_status st = decoder_run(b, *t);
        t->dec_st = st;
        if (st != Q3_OK) t->dec_err = q3_last_error();
        {
            std::lock_guard<std::mutex> g(d.mu);
            t->dec_done.store(true);
        }
        d.cv_done.notify_all();
    }
}
static void stage_join(q3_batcher* b) { if (b->stage.thr.joinable()) b->stage.thr.join(); }
static void stage_drop(q3_batcher* b) {
    stage_join(b);
    if (b->stage.side) { q3_session_free(b->stage.side); b->stage.side = nullptr; }
    b->stage.id = -1; b->stage.st = Q3_OK; b->stage.err.clear();
}

// a row that has nothing to do: frozen on the device from its current frame on (rows of a freshly opened session that no
// request occupies yet)
static q3_status session_idle_row(q3_session* s, int b) {
    SeqInfo& q = s->seq[b];
    int ran = s->frames_run - q.start_run; if (ran < 0) ran = 0; if (ran > q.limit) ran = q.limit;
    q.limit = ran;
    HIPC(sync_frames(s));          // no frame in flight while the row's limit and pages change
    HIPC(q3_hipMemcpy(s->limit + b, &q.limit, sizeof(int), hipMemcpyHostToDevice));
    // A frozen row still runs through every frame (its results are dropped): it reads its keys and rewrites the K/V of its
    // frozen position, prefill_len + ran. It keeps the ONE page that position lies in and every table entry it can reach points
    // there, zero-filled (below); the other pages go back to the pool, and the row takes no
    // more (kv_reserve_frames skips it) — a finished row must not sit on pages the queue is waiting for.
    if (s->paged && !q.idle && !s->kv_rows[(size_t)b].empty()) {
        std::vector<float*>& row = s->kv_rows[(size_t)b];
        size_t keep = (size_t)(q.prefill_len + ran) / KV_PAGE_POS; if (keep >= row.size()) keep = row.size() - 1;
        float* kept = row[keep];
        std::vector<float*> back;
        for (size_t i = 0; i < row.size(); ++i) if (i != keep) back.push_back(row[i]);
        if (!back.empty()) (s->kv_in_bf16 ? s->m->kv_pool16 : s->m->kv_pool).give(back);
        row.assign(1, kept);
        // EVERY entry of the row's table names the kept page (none keeps the address of a page that went back to the pool), and the
        // page is zero-filled: the frozen row goes on reading positions 0 .. pos through it, i.e. also slots this row never wrote
        // (a previous owner's bits, possibly NaN / Inf, which would then run through the sampler and the embedding gather of a row
        // nobody reads); zeros are finite keys. One strided memset per K and V: n_layers runs of nkv * KV_PAGE_POS * HEAD_DIM elements.
        std::vector<unsigned long long> ent((size_t)KV_MAX_PAGES, (unsigned long long)kept);
        HIPC(q3_hipMemcpy(s->kv_table + (size_t)b * KV_MAX_PAGES, ent.data(), ent.size() * 8, hipMemcpyHostToDevice));
        const KvPool& pool = s->kv_in_bf16 ? s->m->kv_pool16 : s->m->kv_pool;
        const size_t run_bytes = pool.run_floats * pool.elem_bytes, pitch = pool.layer_stride() * pool.elem_bytes;
        HIPC(hipMemset2DAsync(kept, pitch, 0, run_bytes, (size_t)pool.n_layers, s->stream));
        HIPC(hipMemset2DAsync((char*)kept + pool.v_delta() * pool.elem_bytes, pitch, 0, run_bytes, (size_t)pool.n_layers, s->stream));
        HIPC(hipStreamSynchronize(s->stream));
    }
    q.idle = true;
    s->codes_host_valid = false;
    return Q3_OK;
}

extern "C" q3_status q3_batcher_create(q3_model* m, int slots, int frame_budget, int prompt_budget, q3_batcher** out) {
    if (!m || !out) return set_err(Q3_INVALID_ARG, "q3_batcher_create: null argument");
    if (!m->finalized) return set_err(Q3_INVALID_ARG, "model not finalized");
    if (slots < 1 || slots > Q3_MAX_BATCH) return set_err(Q3_UNSUPPORTED, "q3_batcher_create: %d rows unsupported (1..%d)", slots, Q3_MAX_BATCH);
    if (frame_budget < 1 || prompt_budget < 0) return set_err(Q3_INVALID_ARG, "q3_batcher_create: frame_budget must be >= 1, prompt_budget >= 0");
    {   // the session the first step opens: max_seq = max(prompt_budget, 16) + frame_budget + 1 positions of the RoPE table
        const long need = (long)(prompt_budget > 16 ? prompt_budget : 16) + frame_budget + 1;
        if (need > m->rope_len) return set_err(Q3_KV_OVERFLOW, "q3_batcher_create: prompt_budget + frame_budget = %ld positions exceed the RoPE table (%d)", need, m->rope_len);
    }
    std::unique_ptr<q3_batcher> b(new q3_batcher());
    b->m = m; b->slots = slots; b->frame_budget = frame_budget; b->prompt_budget = prompt_budget;
    b->owner.assign(slots, -1); b->commit.assign(slots, 0);
    b->dec.reset(new BatDecoder());
    *out = b.release();
    return Q3_OK;
}
extern "C" void q3_batcher_free(q3_batcher* b) {
    if (!b) return;
    stage_drop(b);
    if (b->dec) decoder_stop(b);                       // finishes what is queued (tickets nobody will fetch included), then ends the worker
    if (b->s) q3_session_free(b->s);
    delete b;
}
extern "C" q3_status q3_batcher_submit(q3_batcher* b, const q3_request* req, int want_pcm, int64_t* ticket) {
    if (!b || !req || !ticket) return set_err(Q3_INVALID_ARG, "q3_batcher_submit: null argument");
    if (req->opts.max_length < 1 || req->opts.max_length > b->frame_budget)
        return set_err(Q3_UNSUPPORTED, "q3_batcher_submit: max_length %d outside 1..%d (the batcher's frame budget)", req->opts.max_length, b->frame_budget);
    if (req->n_text < 0 || req->n_instruct < 0 || req->n_ref < 0 || req->n_ref_text < 0) return set_err(Q3_INVALID_ARG, "q3_batcher_submit: negative length");
    std::unique_ptr<BatTicket> t(new BatTicket());
    t->req.own(*req, b->m->cfg.hidden); t->want_pcm = want_pcm != 0;
    const int64_t id = b->next_id++;
    b->t[id] = std::move(t);
    b->queue.push_back(id);
    *ticket = id;
    return Q3_OK;
}

static void bat_fail(BatTicket& t, q3_status st) { t.state = Q3_TICKET_FAILED; t.st = st; t.err = q3_last_error(); t.row = -1; }

// the row's sequence has ended: keep its codes (and PCM), free the row
static q3_status bat_collect(q3_batcher* b, int row) {
    BatTicket& t = *b->t[b->owner[row]];
    int n = 0;
    Q3C(q3_session_codes(b->s, row, nullptr, 0, &n));
    t.codes.resize((size_t)n * 16); t.n_frames = n;
    if (n > 0) Q3C(q3_session_codes(b->s, row, t.codes.data(), n, &n));
    bool async = false;
    if (t.want_pcm && n > 0) {
        static const bool sync_decode = getenv("Q3_BAT_SYNC_DECODE") != nullptr;
        t.pcm.resize((size_t)n * samples_per_frame(b->m->cfg));
        if (!sync_decode && b->s->seq[(size_t)row].ref_codes.empty()) {
            // the worker vocodes it from the host copy of the codes while the row is refilled and the frames go on
            t.decoding = true; t.dec_done.store(false);
            try { decoder_push(b, &t); async = true; } catch (...) { t.decoding = false; }
        }
        if (!async) {
            size_t ns = 0;
            Q3C(q3_session_decode(b->s, row, 0, n, t.pcm.data(), t.pcm.size(), &ns));
            t.pcm.resize(ns);
        }
    }
    t.state = async ? Q3_TICKET_RUNNING : Q3_TICKET_DONE; t.row = -1;
    b->owner[row] = -1; b->commit[row] = 0;
    // the device freezes a row at its frame limit, not at EOS: idle it now so that it stops advancing — and taking pages — while
    // the queue is empty or waits for room; its pages but one go back to the pool
    return session_idle_row(b->s, row);
}

extern "C" q3_status q3_batcher_step(q3_batcher* b, int n_frames, int use_graph, int* n_running, int* n_queued, int* n_finished) {
    if (!b) return set_err(Q3_INVALID_ARG, "q3_batcher_step: null batcher");
    if (n_frames < 1) return set_err(Q3_INVALID_ARG, "q3_batcher_step: n_frames must be >= 1");
    int finished = 0;
    // Open the session on `slots` idle rows: copies of a one-token CustomVoice prompt with a one-frame limit (ten prefill
    // positions per row — opening on the first request itself would prefill, and size every row's KV extent for, `slots`
    // copies of what may be a 4k-token prompt), frozen before the first frame. Every request, the first included, then enters
    // through q3_session_replace, so prompt kinds mix freely.
    if (!b->s && !b->queue.empty()) {
        // The idle rows are built from fixed, known-valid values — never from a queued request: a malformed first request must
        // fail alone, at its own q3_session_replace below, not wedge the queue by failing the session every step.
        const q3_request& first = b->t[b->queue.front()]->req.r;
        b->chunk_frames = first.opts.chunk_frames >= 1 ? first.opts.chunk_frames : 10;       // the one option a session shares
        const q3_request d = idle_request(b->chunk_frames);
        std::vector<q3_request> reqs((size_t)b->slots, d);
        q3_session* s = nullptr;
        q3_status st = session_create(b->m, reqs.data(), b->slots, b->frame_budget, b->prompt_budget > 16 ? b->prompt_budget : 16, &s);
        if (st == Q3_OK) st = q3_session_prefill(s);
        if (st != Q3_OK) {
            // nothing a request could have caused (the budgets were checked at q3_batcher_create): a device failure. The head
            // ticket takes the error so that a serving loop sees it on a ticket and the queue moves on.
            if (s) q3_session_free(s);
            const int64_t id = b->queue.front(); b->queue.erase(b->queue.begin());
            bat_fail(*b->t[id], st);
            if (n_running) *n_running = 0; if (n_queued) *n_queued = (int)b->queue.size(); if (n_finished) *n_finished = 1;
            return st;
        }
        b->s = s;
        for (int r = 0; r < b->slots; ++r) Q3C(session_idle_row(s, r));
    }
    if (!b->s) { if (n_running) *n_running = 0; if (n_queued) *n_queued = 0; if (n_finished) *n_finished = finished; return Q3_OK; }
    // Admission under a page limit (q3_model_kv_pool_limit): a request enters a row only if its WORST CASE (prompt + max_length
    // positions; row_worst_units) fits beside what the running rows may still come to hold and what everything else on the model
    // holds now — so a page shortage shows up here, as a request that waits in the queue (rows are running: room will come) or
    // fails on its ticket (it cannot fit even alone), never in the middle of a generation where it would stop every row.
    auto held_units = [&](int r) -> long { return (long)b->s->kv_rows[(size_t)r].size() * (b->s->kv_in_bf16 ? 1 : 2); };
    auto admit = [&](const q3_request& rq, long* units_out, bool* wait) -> bool {
        *wait = false; *units_out = 0;
        if (!b->s->paged) return true;
        int S = 0, lim = 0; request_shape(rq, &S, &lim);
        const long units = row_worst_units(S, lim, b->s->kv_bf16);
        *units_out = units;
        long mine = 0, claimed = 0; int running = 0;
        for (int r = 0; r < b->slots; ++r) {
            const long h = held_units(r);
            mine += h;
            if (b->owner[r] >= 0) { claimed += std::max(b->commit[r], h); running++; } else claimed += h;
        }
        std::lock_guard<std::mutex> g(b->m->kv_budget.mu);
        if (b->m->kv_budget.limit <= 0) return true;
        const long others = b->m->kv_budget.used - mine;
        if (others + claimed + units <= b->m->kv_budget.limit) return true;
        *wait = running > 0;
        return false;
    };
    auto fill = [&]() -> q3_status {               // free rows <- waiting requests
        for (int r = 0; r < b->slots && !b->queue.empty(); ++r) {
            if (b->owner[r] >= 0) continue;
            while (!b->queue.empty()) {
                const int64_t id = b->queue.front();
                BatTicket& t = *b->t[id];
                long units = 0; bool wait = false;
                if (!admit(t.req.r, &units, &wait)) {
                    if (wait) return Q3_OK;          // FIFO: the head of the queue waits for running rows to end
                    b->queue.erase(b->queue.begin());
                    if (b->stage.id == id) stage_drop(b);      // (a limit set after it was prefilled ahead) its side session goes with it
                    int S = 0, lim = 0; request_shape(t.req.r, &S, &lim);
                    bat_fail(t, set_err(Q3_KV_OVERFLOW, "KV page pool exhausted: the request's %d prompt positions + %d frames need %ld page(s) (f32 equivalents), more than the pool's limit leaves",
                                        S, lim, (units + 1) / 2));
                    finished++; continue;
                }
                b->queue.erase(b->queue.begin());
                t.req.r.opts.chunk_frames = b->chunk_frames;        // the one option a session shares
                q3_status st;
                if (b->stage.id == id) {
                    // prefilled ahead on the worker's stream: wait for it (normally long done), then only the state copy stands
                    // between two frames of the live rows
                    stage_join(b);
                    st = b->stage.st;
                    if (st != Q3_OK) set_err(st, "%s", b->stage.err.c_str());
                    else {
                        st = sync_frames(b->s) == hipSuccess ? Q3_OK : Q3_HIP_ERROR;      // no frame of the host session in flight while its row changes
                        if (st == Q3_OK) st = transplant_row(b->s, r, b->stage.side, 0, b->stage.limit);
                    }
                    stage_drop(b);
                } else
                    st = q3_session_replace(b->s, r, &t.req.r);
                if (st != Q3_OK) { bat_fail(t, st); finished++; continue; }     // does not fit: the ticket carries the reason; try the next one
                t.state = Q3_TICKET_RUNNING; t.row = r; b->owner[r] = id; b->commit[r] = units;
                break;
            }
        }
        return Q3_OK;
    };
    // the head of the queue starts its prefill on the worker (see q3_batcher::Stage); called with frames about to be queued
    auto stage_begin = [&]() {
        static const bool off = getenv("Q3_BAT_NO_STAGE") != nullptr;
        if (off || b->stage.id >= 0 || b->queue.empty() || b->s->debug || b->s->profile) return;
        // not while this thread may still CAPTURE the host session's frame (the first graph step): one thing less to go wrong
        // (captures are in relaxed mode and repeated when invalidated: q3_session.hip frame_capture)
        if (use_graph ? b->s->graph == nullptr : false) return;
        { std::lock_guard<std::mutex> g(b->m->kv_budget.mu); if (b->m->kv_budget.limit > 0) return; }
        bool any_free = false;
        for (int r = 0; r < b->slots; ++r) any_free = any_free || b->owner[r] < 0;
        if (any_free) return;                        // a free row takes the head at once (fill): nothing to run ahead of
        const int64_t id = b->queue.front();
        BatTicket& t = *b->t[id];
        q3_request rq = t.req.r;                     // (arrays owned by the ticket, which lives until it is fetched)
        rq.opts.chunk_frames = b->chunk_frames;
        const int limit_req = rq.opts.max_length;
        if (limit_req < 1 || limit_req > b->s->max_frames) return;      // the synchronous path reports it on the ticket
        rq.opts.max_length = b->s->max_frames;       // the side session draws the row's PCG stream with the host session's stride
        b->stage.id = id; b->stage.st = Q3_OK; b->stage.err.clear(); b->stage.side = nullptr; b->stage.limit = 0;
        q3_batcher* bp = b;
        try {
        b->stage.thr = std::thread([bp, rq, limit_req]() {
            q3_batcher::Stage& g = bp->stage;
            q3_session* side = nullptr;
            q3_status st = hipSetDevice(bp->m->device) == hipSuccess ? Q3_OK : set_err(Q3_HIP_ERROR, "hipSetDevice");
            if (st == Q3_OK) st = session_create(bp->m, &rq, 1, 0, 0, &side);          // a stream of its own: runs beside the frames
            if (st == Q3_OK) { side->kv_bf16 = bp->s->kv_bf16; st = transplant_check(bp->s, side, 0, limit_req, &g.limit); }
            if (st == Q3_OK) st = q3_session_prefill(side);                            // ends with a synchronisation of that stream
            g.side = side; g.st = st;
            if (st != Q3_OK) g.err = q3_last_error();
        });
        } catch (...) { b->stage.id = -1; }          // no thread to be had: the swap prefills synchronously, as before
    };
    // Run in pieces that end where the next row reaches its frame limit: that row is collected and refilled at once instead of
    // idling to the end of the step (a row that ends on EOS is noticed at q3_session_generate's 32-frame check or at the
    // end of the piece)
    for (int left = n_frames; left > 0;) {
        Q3C(fill());
        int piece = left, busy = 0;
        for (int r = 0; r < b->slots; ++r) {
            if (b->owner[r] < 0) continue;
            const SeqInfo& q = b->s->seq[r];
            const int rem = q.limit - (b->s->frames_run - q.start_run);
            if (rem > 0) { busy++; if (rem < piece) piece = rem; }
        }
        if (busy > 0) {
            stage_begin();
            const q3_status gst = q3_session_generate(b->s, piece, use_graph);
            if (gst == Q3_KV_OVERFLOW && b->s->kv_overflow_row >= 0 && b->owner[b->s->kv_overflow_row] >= 0) {
                // (only reachable when something outside this batcher took the pages its admission counted on) nothing ran: the
                // row that needs the page fails alone and is frozen; the others go on
                const int row = b->s->kv_overflow_row;
                bat_fail(*b->t[b->owner[row]], gst);
                b->owner[row] = -1; b->commit[row] = 0;
                Q3C(session_idle_row(b->s, row));
                finished++;
                continue;
            }
            Q3C(gst);
            left -= piece;
        }
        int collected = 0;
        for (int r = 0; r < b->slots; ++r) {
            if (b->owner[r] < 0) continue;
            {   // a row without a live EOS id ends exactly at its frame limit, which the host knows: no device read-back (a
                // synchronisation and nine blocking copies per step) while no row can have ended
                const SeqInfo& q = b->s->seq[r];
                if (q.req.opts.eos_token_id < 0 && b->s->frames_run - q.start_run < q.limit) continue;
            }
            int n = 0, done = 0;
            Q3C(q3_session_frames(b->s, r, &n, &done));
            if (done) { Q3C(bat_collect(b, r)); finished++; collected++; }
        }
        if (busy == 0 && collected == 0) break;          // nothing runs and nothing is waiting for a row
    }
    Q3C(fill());                                   // the next step starts with full rows
    int running = 0;
    for (int r = 0; r < b->slots; ++r) running += b->owner[r] >= 0 ? 1 : 0;
    // Nothing runs and nothing waits: the rows that just ended may still be with the decode worker. Their samples are waited for
    // HERE, so that a host loop that stops on "running == 0 && queued == 0" finds every ticket DONE, as it always did.
    if (running == 0 && b->queue.empty())
        for (auto& kv : b->t) ticket_wait_decode(b, *kv.second);
    if (n_running) *n_running = running;
    if (n_queued) *n_queued = (int)b->queue.size();
    if (n_finished) *n_finished = finished;
    return Q3_OK;
}

extern "C" q3_status q3_batcher_poll(q3_batcher* b, int64_t ticket, int* state, int* n_frames, size_t* n_samples) {
    if (!b) return set_err(Q3_INVALID_ARG, "q3_batcher_poll: null batcher");
    auto it = b->t.find(ticket);
    if (it == b->t.end()) return set_err(Q3_INVALID_ARG, "q3_batcher_poll: unknown ticket %lld", (long long)ticket);
    BatTicket& t = *it->second;
    if (t.decoding && t.dec_done.load()) ticket_wait_decode(b, t);      // its samples have landed: DONE (or FAILED) from here on
    if (state) *state = t.state;                                         // a ticket still being vocoded reads RUNNING
    int nf = t.n_frames;
    if (t.state == Q3_TICKET_RUNNING && b->s && t.row >= 0) {       // frames run so far (an EOS inside them is only looked at when the row is collected)
        const SeqInfo& q = b->s->seq[t.row];
        nf = b->s->frames_run - q.start_run; if (nf > q.limit) nf = q.limit; if (nf < 0) nf = 0;
    }
    if (n_frames) *n_frames = nf;
    if (n_samples) *n_samples = t.pcm.size();         // (of a ticket still being vocoded: the samples it WILL hold — the size q3_batcher_fetch wants)
    return Q3_OK;
}

extern "C" q3_status q3_batcher_fetch(q3_batcher* b, int64_t ticket, uint32_t* codes_host, int cap_frames, float* pcm_host, size_t cap_samples) {
    if (!b) return set_err(Q3_INVALID_ARG, "q3_batcher_fetch: null batcher");
    auto it = b->t.find(ticket);
    if (it == b->t.end()) return set_err(Q3_INVALID_ARG, "q3_batcher_fetch: unknown ticket %lld", (long long)ticket);
    BatTicket& t = *it->second;
    ticket_wait_decode(b, t);                         // a row that ended but is still being vocoded: wait for its samples
    if (t.state == Q3_TICKET_FAILED) {
        const q3_status st = t.st; const std::string err = t.err;
        b->t.erase(it);
        return set_err(st, "%s", err.c_str());
    }
    if (t.state != Q3_TICKET_DONE) return set_err(Q3_INVALID_ARG, "q3_batcher_fetch: ticket %lld has not finished", (long long)ticket);
    if (codes_host) {
        if (cap_frames < t.n_frames) return set_err(Q3_INVALID_ARG, "codes buffer too small (%d < %d frames)", cap_frames, t.n_frames);
        memcpy(codes_host, t.codes.data(), t.codes.size() * 4);
    }
    if (pcm_host) {
        if (cap_samples < t.pcm.size()) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
        memcpy(pcm_host, t.pcm.data(), t.pcm.size() * 4);
    }
    b->t.erase(it);
    return Q3_OK;
}

