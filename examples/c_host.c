/* c_host.c — a host WITHOUT Python or torch driving libq3tts.so through the C ABI only (include/q3tts.h): what the
 * reference's Rust host does through its `extern "C"` block, written in C because this image has no
 * Rust toolchain. Mirrors examples/tts.rs of the reference: load (here: seeded synthetic weights, every tensor filled
 * with q3_synth_fill scale 0.04 — no checkpoint files exist offline), one data-parallel rank (RCCL communicator of
 * world size 1: unique id -> init -> weight broadcast), synthesize one utterance with default sampling, write
 * <out>/c_host.wav + <out>/c_host_codes.bin.
 *
 *   gcc -O1 -Iinclude examples/c_host.c -o build/c_host -Lqwen3_tts_rs_amd -lq3tts -Wl,-rpath,$PWD/qwen3_tts_rs_amd
 *   build/c_host <out_dir> [frames]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "q3tts.h"

#define CHECK(expr)                                                                              \
    do {                                                                                         \
        q3_status s_ = (expr);                                                                   \
        if (s_ != Q3_OK) { fprintf(stderr, "%s failed (%d): %s\n", #expr, (int)s_, q3_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const char* out_dir = argc > 1 ? argv[1] : ".";
    const int frames = argc > 2 ? atoi(argv[2]) : 6;
    if (q3_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }

    /* a small talker / code predictor in front of the full-size 12 Hz decoder */
    q3_config cfg;
    CHECK(q3_config_default(0, &cfg));
    cfg.text_vocab = 512; cfg.text_dim = 128; cfg.hidden = 128; cfg.inter = 256; cfg.n_layers = 2;
    cfg.n_heads = 2; cfg.n_kv_heads = 1; cfg.cp_hidden = 128; cfg.cp_inter = 256; cfg.cp_layers = 2; cfg.cp_heads = 2; cfg.cp_kv_heads = 1;

    q3_model* model = NULL;
    CHECK(q3_model_create(&cfg, 0, &model));
    const int nt = q3_model_n_tensors(model);
    for (int i = 0; i < nt; ++i) {
        const char* name; int64_t n; int stored;
        CHECK(q3_model_tensor_info(model, i, &name, &n, &stored));
        void* buf = malloc((size_t)n * 4);
        const int is_norm = strstr(name, "norm.weight") != NULL || strstr(name, "cluster_usage") != NULL;
        CHECK(q3_synth_fill(7, name, stored, is_norm ? 0.01f : 0.04f, is_norm ? 1.0f : 0.0f, n, buf));
        CHECK(q3_model_set_tensor(model, name, stored, buf, n));
        free(buf);
    }

    /* data-parallel rank 0 of 1: the same three calls every rank of an N-GPU job makes */
    unsigned char id[Q3_DP_ID_BYTES];
    q3_dp_comm* comm = NULL;
    CHECK(q3_dp_unique_id(id));
    CHECK(q3_dp_init(0, 1, id, 0, &comm));
    CHECK(q3_dp_broadcast_weights(comm, model, 0));
    CHECK(q3_model_finalize(model));

    uint32_t text[12];
    for (int i = 0; i < 12; ++i) text[i] = (uint32_t)(17 * i + 3) % 512;
    q3_request req;
    memset(&req, 0, sizeof req);
    req.mode = Q3_MODE_CUSTOM_VOICE; req.text_ids = text; req.n_text = 12;
    req.speaker_id = 3061; req.language_id = 2050;           /* Speaker::Ryan, Language::English (talker.rs:94-157) */
    req.opts.temperature = 0.9; req.opts.top_p = 0.9; req.opts.repetition_penalty = 1.05; req.opts.top_k = 50;
    req.opts.seed = 42; req.opts.has_seed = 1; req.opts.max_length = frames; req.opts.eos_token_id = -1;
    req.opts.chunk_frames = 10; req.opts.min_new_tokens = 2;

    q3_session* sess = NULL;
    CHECK(q3_session_create(model, &req, 1, &sess));
    const size_t cap = (size_t)frames * 1920;
    float* pcm = (float*)malloc(cap * sizeof(float));
    size_t n_samples = 0; q3_timing tm;
    float* outs[1] = {pcm};
    CHECK(q3_session_run(sess, 1, outs, &cap, &n_samples, &tm));
    uint32_t* codes = (uint32_t*)malloc((size_t)frames * 16 * 4);
    int nf = 0;
    CHECK(q3_session_codes(sess, 0, codes, frames, &nf));

    double mine[2] = {tm.generation_ms, (double)nf}, all[2];
    CHECK(q3_dp_allgather_f64(comm, mine, 2, all));
    char path[1024];
    snprintf(path, sizeof path, "%s/c_host.wav", out_dir);
    CHECK(q3_wav_write_pcm16(path, pcm, (int64_t)n_samples, 24000));
    snprintf(path, sizeof path, "%s/c_host_codes.bin", out_dir);
    CHECK(q3_codes_write_bin(path, codes, nf, 16));
    printf("frames %d samples %zu prefill %.2f ms generation %.2f ms decode %.2f ms (gathered: %.2f ms, %.0f frames)\n", nf, n_samples,
           tm.prefill_ms, tm.generation_ms, tm.decode_ms, all[0], all[1]);
    /* the same request and a second one through the native continuous batcher (two rows): the library's own serving loop;
     * the first ticket must come back with exactly the codes of the session above */
    q3_batcher* bat = NULL;
    CHECK(q3_batcher_create(model, 2, frames, 0, &bat));
    uint32_t text2[7];
    for (int i = 0; i < 7; ++i) text2[i] = (uint32_t)(29 * i + 5) % 512;
    q3_request req2 = req;
    req2.text_ids = text2; req2.n_text = 7; req2.opts.seed = 7; req2.opts.max_length = frames > 2 ? frames - 2 : frames;
    int64_t ta = 0, tb = 0;
    CHECK(q3_batcher_submit(bat, &req, 0, &ta));
    CHECK(q3_batcher_submit(bat, &req2, 1, &tb));
    for (int guard = 0; guard < 1000; ++guard) {
        int running = 0, queued = 0, fin = 0;
        CHECK(q3_batcher_step(bat, 4, 1, &running, &queued, &fin));
        if (running == 0 && queued == 0) break;
    }
    int st_a = 0, st_b = 0, nfa = 0, nfb = 0; size_t nsa = 0, nsb = 0;
    CHECK(q3_batcher_poll(bat, ta, &st_a, &nfa, &nsa));
    CHECK(q3_batcher_poll(bat, tb, &st_b, &nfb, &nsb));
    if (st_a != Q3_TICKET_DONE || st_b != Q3_TICKET_DONE || nfa != nf || nsb != (size_t)nfb * 1920) { fprintf(stderr, "batcher: unexpected ticket state\n"); return 3; }
    uint32_t* codes_a = (uint32_t*)malloc((size_t)nfa * 16 * 4);
    float* pcm_b = (float*)malloc(nsb * sizeof(float));
    CHECK(q3_batcher_fetch(bat, ta, codes_a, nfa, NULL, 0));
    CHECK(q3_batcher_fetch(bat, tb, NULL, 0, pcm_b, nsb));
    if (memcmp(codes_a, codes, (size_t)nf * 16 * 4) != 0) { fprintf(stderr, "batcher: codes differ from the session's\n"); return 3; }
    printf("batcher: 2 requests through 2 rows, %d + %d frames, ticket 1 bit-equal to the session\n", nfa, nfb);
    q3_batcher_free(bat); free(codes_a); free(pcm_b);

    q3_session_free(sess); q3_dp_free(comm); q3_model_free(model);
    free(pcm); free(codes);
    return 0;
}
