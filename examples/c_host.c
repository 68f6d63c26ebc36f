/* c_host.c — a host WITHOUT Python or torch driving libq3tts.so through the C ABI only (include/q3tts.h): what the
 * reference's Rust host does through its `extern "C"` block, written in C because this image has no
 * Rust toolchain. Mirrors examples/tts.rs of the reference: load (here: seeded synthetic weights, every tensor filled
 * with q3_synth_fill scale 0.04 — no checkpoint files exist offline), one data-parallel rank (RCCL communicator of
 * world size 1: unique id -> init -> weight broadcast), synthesize one utterance with default sampling, write
 * <out>/c_host.wav + <out>/c_host_codes.bin.
 *
 *   gcc -O1 -Iinclude examples/c_host.c -o build/c_host -Lqwen3_tts_rs_amd -lq3tts -Wl,-rpath,$PWD/qwen3_tts_rs_amd
 *   build/c_host <out_dir> [frames]
 *
 * As rank r of an N-GPU job (tools/run_c_host_ranks.sh starts the N processes): Q3_RANK=r Q3_WORLD=N Q3_ID_FILE=<path>
 * [Q3_DEVICE=d, default r]. Rank 0 creates the RCCL unique id and publishes it through the file (written under a temporary
 * name, then renamed: a reader never sees half an id); only rank 0 fills the weights — the others receive the arena through
 * the one broadcast, then q3_model_mark_loaded + q3_model_finalize. Every rank synthesises its own utterance
 * (seed 42 + rank) and the ranks exchange their timings with q3_dp_allgather_f64; rank 0 prints the job's frames/s.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
    const int rank = getenv("Q3_RANK") ? atoi(getenv("Q3_RANK")) : 0, world = getenv("Q3_WORLD") ? atoi(getenv("Q3_WORLD")) : 1;
    const int device = getenv("Q3_DEVICE") ? atoi(getenv("Q3_DEVICE")) : (world > 1 ? rank : 0);
    const char* id_file = getenv("Q3_ID_FILE");
    if (world < 1 || world > 64 || rank < 0 || rank >= world || (world > 1 && !id_file)) { fprintf(stderr, "bad Q3_RANK / Q3_WORLD / Q3_ID_FILE\n"); return 2; }
    if (device >= q3_device_count()) { fprintf(stderr, "rank %d: device %d of %d\n", rank, device, q3_device_count()); return 2; }

    /* a small talker / code predictor in front of the full-size 12 Hz decoder */
    q3_config cfg;
    CHECK(q3_config_default(0, &cfg));
    cfg.text_vocab = 512; cfg.text_dim = 128; cfg.hidden = 128; cfg.inter = 256; cfg.n_layers = 2;
    cfg.n_heads = 2; cfg.n_kv_heads = 1; cfg.cp_hidden = 128; cfg.cp_inter = 256; cfg.cp_layers = 2; cfg.cp_heads = 2; cfg.cp_kv_heads = 1;

    q3_model* model = NULL;
    CHECK(q3_model_create(&cfg, device, &model));
    const int nt = q3_model_n_tensors(model);
    for (int i = 0; i < nt && rank == 0; ++i) {      /* the checkpoint exists on rank 0 only; the others get it by broadcast */
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
    if (rank == 0) {
        CHECK(q3_dp_unique_id(id));
        if (world > 1) {
            char tmp[1024]; snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, id_file) != 0) { fprintf(stderr, "cannot publish the id\n"); return 2; }
        }
    } else {
        int got = 0;
        for (int tries = 0; tries < 6000 && !got; ++tries) {       /* up to 60 s for rank 0 to come up */
            FILE* f = fopen(id_file, "rb");
            if (f) { got = fread(id, 1, sizeof id, f) == sizeof id; fclose(f); }
            if (!got) usleep(10000);
        }
        if (!got) { fprintf(stderr, "rank %d: no id in %s\n", rank, id_file); return 2; }
    }
    CHECK(q3_dp_init(rank, world, id, device, &comm));
    CHECK(q3_dp_broadcast_weights(comm, model, 0));
    if (rank != 0) CHECK(q3_model_mark_loaded(model));
    CHECK(q3_model_finalize(model));

    uint32_t text[12];
    for (int i = 0; i < 12; ++i) text[i] = (uint32_t)(17 * i + 3) % 512;
    q3_request req;
    memset(&req, 0, sizeof req);
    req.mode = Q3_MODE_CUSTOM_VOICE; req.text_ids = text; req.n_text = 12;
    req.speaker_id = 3061; req.language_id = 2050;           /* Speaker::Ryan, Language::English (talker.rs:94-157) */
    req.opts.temperature = 0.9; req.opts.top_p = 0.9; req.opts.repetition_penalty = 1.05; req.opts.top_k = 50;
    req.opts.seed = 42 + (uint64_t)rank; req.opts.has_seed = 1; req.opts.max_length = frames; req.opts.eos_token_id = -1;
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

    double mine[2] = {tm.prefill_ms + tm.generation_ms + tm.decode_ms, (double)nf}, all[2 * 64];
    CHECK(q3_dp_allgather_f64(comm, mine, 2, all));
    char path[1024];
    if (world > 1) snprintf(path, sizeof path, "%s/c_host_rank%d.wav", out_dir, rank); else snprintf(path, sizeof path, "%s/c_host.wav", out_dir);
    CHECK(q3_wav_write_pcm16(path, pcm, (int64_t)n_samples, 24000));
    if (world > 1) snprintf(path, sizeof path, "%s/c_host_rank%d_codes.bin", out_dir, rank); else snprintf(path, sizeof path, "%s/c_host_codes.bin", out_dir);
    CHECK(q3_codes_write_bin(path, codes, nf, 16));
    printf("frames %d samples %zu prefill %.2f ms generation %.2f ms decode %.2f ms (gathered: %.2f ms, %.0f frames)\n", nf, n_samples,
           tm.prefill_ms, tm.generation_ms, tm.decode_ms, all[0], all[1]);
    if (world > 1 && rank == 0) {           /* the job: frames of all ranks / the slowest rank's time */
        double slowest = 0, total = 0;
        for (int r = 0; r < world; ++r) { if (all[2 * r] > slowest) slowest = all[2 * r]; total += all[2 * r + 1]; }
        printf("job: %d ranks, %.0f frames, slowest rank %.2f ms -> %.1f frames/s\n", world, total, slowest, total / (slowest / 1e3));
    }
    if (world > 1) { q3_session_free(sess); q3_dp_free(comm); q3_model_free(model); free(pcm); free(codes); return 0; }
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
