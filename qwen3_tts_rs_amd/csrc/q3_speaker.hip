// q3_speaker.hip — speaker-embedding path of x-vector voice cloning on gfx950: 24 kHz reference audio -> log-mel
// (audio/mel.rs:47-59, 135-227) -> ECAPA-TDNN (models/speaker.rs:345-469) -> [enc_dim] embedding.
//
// Once-per-utterance work (≈9 GFLOP for 5 s of reference audio), so the design goal is "no host arithmetic, few
// moving parts": every convolution goes through the vocoder's bf16x3 matrix-core conv (q3_kernels_codec.hip,
// f32-equivalent accuracy) — 1x1 convs directly, k > 1 "same" convs as a causal conv over a reflect-padded copy
// whose first (k-1)·dil output columns are dropped — plus five small kernels: STFT+mel (DFT in f64: MI355X has
// the FP64 rate to make an FFT pointless at 1024 points), reflect-pad/add, column copy, SE scale+residual, and the
// attentive-statistics reductions.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "q3_internal.h"
#include "q3_capture_lock.h"
#include "q3_kernels.h"

namespace q3 {

__device__ __forceinline__ float block_sum(float v, float* sh) {      // 256 threads
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// One workgroup per STFT frame (mel.rs:168-227 stft, 135-166 compute_for_speaker_encoder): reflect-padded,
// Hann-windowed 1024-point DFT in f64, magnitude sqrt(re² + im² + 1e-9) in f32, mel filterbank, ln(max(., 1e-5)).
__global__ __launch_bounds__(256) void k_spk_stft_mel(const float* __restrict__ x, long n, int T, const float* __restrict__ win,
                                                      const double* __restrict__ cs, const double* __restrict__ sn,
                                                      const float* __restrict__ fb, float* __restrict__ mel, int n_mels) {
    constexpr int NF = 1024, HOP = 256, PAD = (NF - HOP) / 2, NB = NF / 2 + 1;
    __shared__ float buf[NF];
    __shared__ float mag[NB + 7];
    __shared__ double tc[NF], ts[NF];
    const int f = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < NF; j += 256) {
        const long p = (long)f * HOP + j - PAD;
        long idx = p;
        if (p < 0) idx = (-p < n) ? -p : n - 1;                       // mel.rs:176-183
        else if (p >= n) { const long i = p - n; idx = (n >= 2 + i) ? n - 2 - i : 0; }   // mel.rs:185-192
        buf[j] = mul_rn(x[idx], win[j]);
        tc[j] = cs[j]; ts[j] = sn[j];
    }
    __syncthreads();
    for (int k = tid; k < NB; k += 256) {
        double re = 0.0, im = 0.0;
        for (int j = 0; j < NF; ++j) {
            const int idx = (j * k) & (NF - 1);
            const double v = (double)buf[j];
            re += v * tc[idx]; im -= v * ts[idx];
        }
        const float rf = (float)re, jf = (float)im;
        mag[k] = sqrtf(add_rn(add_rn(mul_rn(rf, rf), mul_rn(jf, jf)), 1e-9f));
    }
    __syncthreads();
    for (int m = tid; m < n_mels; m += 256) {
        const float* row = fb + (size_t)m * NB;
        float acc = 0.0f;
        for (int k = 0; k < NB; ++k) acc = add_rn(acc, mul_rn(row[k], mag[k]));
        mel[(size_t)m * T + f] = logf(fmaxf(acc, 1e-5f));
    }
}

// out[c][tp] = a[c][r] (+ b[c][r]), r = reflect(tp - pl) (speaker.rs:24-51; the Res2Net "chunk + previous output",
// speaker.rs:186-192, rides along)
__global__ __launch_bounds__(256) void k_spk_pad(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                 int T, int pl, int Tp) {
    const int c = blockIdx.y, tp = blockIdx.x * 256 + threadIdx.x;
    if (tp >= Tp) return;
    int r = tp - pl;
    if (r < 0) r = -r; else if (r >= T) r = 2 * T - 2 - r;
    float v = a[(size_t)c * T + r];
    if (b) v = v + b[(size_t)c * T + r];
    out[(size_t)c * Tp + tp] = v;
}
// dst[c][t] = src[c][off + t]
__global__ __launch_bounds__(256) void k_spk_cols(const float* __restrict__ src, int ld, int off, float* __restrict__ dst, int T) {
    const int c = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t < T) dst[(size_t)c * T + t] = src[(size_t)c * ld + off + t];
}
// per-channel mean over time (SE squeeze, speaker.rs:220)
__global__ __launch_bounds__(256) void k_spk_mean(const float* __restrict__ x, float* __restrict__ mean, int T) {
    __shared__ float sh[4];
    const float* r = x + (size_t)blockIdx.x * T;
    float a = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) a += r[t];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) mean[blockIdx.x] = a / (float)T;
}
// h = o * g[c] + h (SE excite + the block's residual, speaker.rs:224-226, 270-271)
__global__ __launch_bounds__(256) void k_spk_se_apply(const float* __restrict__ o, const float* __restrict__ g, float* __restrict__ h, int T) {
    const int c = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t < T) { const size_t e = (size_t)c * T + t; h[e] = add_rn(mul_rn(o[e], g[c]), h[e]); }
}
// attention input of the pooling (speaker.rs:303-316): rows [x ; mean ; std] with std = sqrt(mean((x-mean)²) + 1e-5)
__global__ __launch_bounds__(256) void k_spk_asp_in(const float* __restrict__ m, float* __restrict__ ain, int C, int T) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    const float* r = m + (size_t)c * T;
    float a = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) a += r[t];
    const float mean = block_sum(a, sh) / (float)T;
    float q = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) { const float d = r[t] - mean; q += d * d; }
    const float sd = sqrtf(block_sum(q, sh) / (float)T + 1e-5f);
    for (int t = threadIdx.x; t < T; t += 256) {
        ain[(size_t)c * T + t] = r[t];
        ain[(size_t)(C + c) * T + t] = mean;
        ain[(size_t)(2 * C + c) * T + t] = sd;
    }
}
// softmax over time of the attention logits, weighted mean and std (speaker.rs:322-342): pooled = [mean ; std]
__global__ __launch_bounds__(256) void k_spk_asp_pool(const float* __restrict__ aw, const float* __restrict__ m, float* __restrict__ pooled, int C, int T) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    const float* a = aw + (size_t)c * T; const float* r = m + (size_t)c * T;
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) mx = fmaxf(mx, a[t]);
    mx = block_max(mx, sh);
    float s = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) s += expf(a[t] - mx);
    s = block_sum(s, sh);
    float wm = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) wm += r[t] * (expf(a[t] - mx) / s);
    wm = block_sum(wm, sh);
    float wv = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) { const float d = r[t] - wm; wv += d * d * (expf(a[t] - mx) / s); }
    wv = block_sum(wv, sh);
    if (threadIdx.x == 0) { pooled[c] = wm; pooled[C + c] = sqrtf(wv + 1e-5f); }
}

}  // namespace q3

using namespace q3;

namespace {
struct SpkSlot { std::string name; int64_t n = 0; size_t offset = 0; bool loaded = false; };
struct SpkConv { const float* w = nullptr; const float* b = nullptr; const void* wpk = nullptr; int cin = 0, cout = 0, k = 1, dil = 1; };
struct SpkBlock { SpkConv tdnn1, tdnn2, se1, se2; std::vector<SpkConv> res; };
std::string fmts(const char* f, int a, int b = 0) { char buf[128]; snprintf(buf, sizeof buf, f, a, b); return buf; }
}  // namespace

struct q3_speaker_encoder {
    q3_spk_config cfg{};
    int device = 0;
    std::vector<SpkSlot> slots;
    std::unordered_map<std::string, int> index;
    char* arena = nullptr; size_t arena_bytes = 0;
    void* wpk_arena = nullptr;
    float* win = nullptr; double* dft_cs = nullptr; double* dft_sn = nullptr; float* fb = nullptr;
    bool finalized = false;
    hipStream_t st = nullptr;
    SpkConv c0, mfa, asp_tdnn, asp_conv, fc;
    SpkBlock blk[3];
    float* ws = nullptr; size_t ws_floats = 0;
};

#define SPK_HIP(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return q3i_set_err(Q3_HIP_ERROR, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static void spk_add(q3_speaker_encoder* e, const std::string& name, int64_t n) {
    SpkSlot s; s.name = name; s.n = n;
    s.offset = (e->arena_bytes + 255) & ~(size_t)255;
    e->arena_bytes = s.offset + (size_t)n * 4;
    e->index[name] = (int)e->slots.size();
    e->slots.push_back(s);
}
static void spk_add_conv(q3_speaker_encoder* e, const std::string& prefix, int cout, int cin, int k) {
    spk_add(e, prefix + ".weight", (int64_t)cout * cin * k);
    spk_add(e, prefix + ".bias", cout);
}

extern "C" q3_status q3_spk_config_default(q3_spk_config* out) {
    if (!out) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_config_default: null");
    q3_spk_config c{};
    c.mel_dim = 128; c.enc_dim = 1024;
    const int ch[5] = {512, 512, 512, 512, 1536}, ks[5] = {5, 3, 3, 3, 1}, dl[5] = {1, 2, 3, 4, 1};
    for (int i = 0; i < 5; ++i) { c.channels[i] = ch[i]; c.kernel_sizes[i] = ks[i]; c.dilations[i] = dl[i]; }
    c.attention_channels = 128; c.res2net_scale = 8; c.se_channels = 128; c.sample_rate = 24000;
    *out = c;
    return Q3_OK;
}

extern "C" q3_status q3_spk_create(const q3_spk_config* cfg, int device, q3_speaker_encoder** out) {
    if (!cfg || !out) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_create: null argument");
    const q3_spk_config& c = *cfg;
    if (c.mel_dim < 1 || c.enc_dim < 1 || c.res2net_scale < 2 || c.se_channels < 1 || c.attention_channels < 1)
        return q3i_set_err(Q3_INVALID_ARG, "q3_spk_create: bad config");
    for (int i = 0; i < 5; ++i)
        if (c.channels[i] < 1 || c.kernel_sizes[i] < 1 || c.dilations[i] < 1) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_create: bad block config");
    for (int i = 1; i <= 3; ++i) {
        if (c.channels[i] != c.channels[0]) return q3i_set_err(Q3_UNSUPPORTED, "SE-Res2Net blocks add their input: channel counts must match");
        if (c.channels[i] % c.res2net_scale) return q3i_set_err(Q3_INVALID_ARG, "channels must divide by res2net_scale");
    }
    if (c.mel_dim != 128 || c.sample_rate != 24000) return q3i_set_err(Q3_UNSUPPORTED, "mel front end is the reference's fixed 24 kHz / 128-band one (mel.rs:47-59)");
    auto* e = new q3_speaker_encoder();
    e->cfg = c; e->device = device;
    spk_add_conv(e, "speaker_encoder.blocks.0.conv", c.channels[0], c.mel_dim, c.kernel_sizes[0]);
    for (int bi = 1; bi <= 3; ++bi) {
        const int C = c.channels[bi], chn = C / c.res2net_scale;
        spk_add_conv(e, fmts("speaker_encoder.blocks.%d.tdnn1.conv", bi), C, C, 1);
        for (int i = 0; i < c.res2net_scale - 1; ++i) spk_add_conv(e, fmts("speaker_encoder.blocks.%d.res2net_block.blocks.%d.conv", bi, i), chn, chn, c.kernel_sizes[bi]);
        spk_add_conv(e, fmts("speaker_encoder.blocks.%d.tdnn2.conv", bi), C, C, 1);
        spk_add_conv(e, fmts("speaker_encoder.blocks.%d.se_block.conv1", bi), c.se_channels, C, 1);
        spk_add_conv(e, fmts("speaker_encoder.blocks.%d.se_block.conv2", bi), C, c.se_channels, 1);
    }
    const int mfa_in = c.channels[1] + c.channels[2] + c.channels[3], C4 = c.channels[4];
    spk_add_conv(e, "speaker_encoder.mfa.conv", C4, mfa_in, c.kernel_sizes[4]);
    spk_add_conv(e, "speaker_encoder.asp.tdnn.conv", c.attention_channels, 3 * C4, 1);
    spk_add_conv(e, "speaker_encoder.asp.conv", C4, c.attention_channels, 1);
    spk_add_conv(e, "speaker_encoder.fc", c.enc_dim, 2 * C4, 1);
    if (device >= 0) {
        hipError_t he = hipSetDevice(device);
        if (he == hipSuccess) he = hipMalloc((void**)&e->arena, e->arena_bytes);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking);
        if (he != hipSuccess) { q3_spk_free(e); return q3i_set_err(Q3_HIP_ERROR, "q3_spk_create: %s", hipGetErrorString(he)); }
    }
    *out = e;
    return Q3_OK;
}

extern "C" void q3_spk_free(q3_speaker_encoder* e) {
    if (!e) return;
    if (e->device >= 0) {
        (void)hipSetDevice(e->device);
        if (e->st) { (void)hipStreamSynchronize(e->st); (void)hipStreamDestroy(e->st); }
        (void)hipFree(e->arena); (void)hipFree(e->wpk_arena); (void)hipFree(e->win); (void)hipFree(e->dft_cs);
        (void)hipFree(e->dft_sn); (void)hipFree(e->fb); (void)hipFree(e->ws);
    }
    delete e;
}

extern "C" q3_status q3_spk_get_config(const q3_speaker_encoder* e, q3_spk_config* out) {
    if (!e || !out) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_get_config: null");
    *out = e->cfg;
    return Q3_OK;
}
extern "C" int q3_spk_n_tensors(const q3_speaker_encoder* e) { return e ? (int)e->slots.size() : 0; }
extern "C" q3_status q3_spk_tensor_info(const q3_speaker_encoder* e, int i, const char** name, int64_t* n) {
    if (!e || i < 0 || i >= (int)e->slots.size()) return q3i_set_err(Q3_INVALID_ARG, "speaker tensor index out of range");
    if (name) *name = e->slots[i].name.c_str();
    if (n) *n = e->slots[i].n;
    return Q3_OK;
}

extern "C" q3_status q3_spk_set_tensor(q3_speaker_encoder* e, const char* name, const void* data, int src_dtype, int64_t n) {
    if (!e || !name || !data) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_set_tensor: null argument");
    if (e->device < 0) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_set_tensor: manifest-only handle");
    auto it = e->index.find(name);
    if (it == e->index.end()) return q3i_set_err(Q3_INVALID_ARG, "unknown speaker-encoder tensor %s", name);
    SpkSlot& s = e->slots[it->second];
    if (s.n != n) return q3i_set_err(Q3_INVALID_ARG, "tensor %s: expected %lld elements, got %lld", name, (long long)s.n, (long long)n);
    SPK_HIP(hipSetDevice(e->device));
    if (src_dtype == Q3_DTYPE_F32) {
        SPK_HIP(q3_hipMemcpy(e->arena + s.offset, data, (size_t)n * 4, hipMemcpyHostToDevice));
    } else if (src_dtype == Q3_DTYPE_BF16) {
        std::vector<float> tmp((size_t)n);
        const uint16_t* h = (const uint16_t*)data;
        for (int64_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)h[i] << 16; memcpy(&tmp[(size_t)i], &u, 4); }
        SPK_HIP(q3_hipMemcpy(e->arena + s.offset, tmp.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    } else {
        return q3i_set_err(Q3_INVALID_ARG, "q3_spk_set_tensor: unsupported source dtype %d", src_dtype);
    }
    s.loaded = true; e->finalized = false;
    return Q3_OK;
}

static SpkConv spk_conv(q3_speaker_encoder* e, const std::string& prefix, int cout, int cin, int k, int dil) {
    SpkConv c; c.cout = cout; c.cin = cin; c.k = k; c.dil = dil;
    c.w = (const float*)(e->arena + e->slots[e->index[prefix + ".weight"]].offset);
    c.b = (const float*)(e->arena + e->slots[e->index[prefix + ".bias"]].offset);
    return c;
}

// Slaney mel scale and filterbank in f32, exactly as mel.rs:243-318 writes them
static float spk_hz_to_mel(float f) {
    const float F_SP = 200.0f / 3.0f, MIN_LOG_HZ = 1000.0f, MIN_LOG_MEL = MIN_LOG_HZ / F_SP, LOGSTEP = 0.06875174f;
    return f < MIN_LOG_HZ ? f / F_SP : MIN_LOG_MEL + logf(f / MIN_LOG_HZ) / LOGSTEP;
}
static float spk_mel_to_hz(float m) {
    const float F_SP = 200.0f / 3.0f, MIN_LOG_HZ = 1000.0f, MIN_LOG_MEL = MIN_LOG_HZ / F_SP, LOGSTEP = 0.06875174f;
    return m < MIN_LOG_MEL ? m * F_SP : MIN_LOG_HZ * expf((m - MIN_LOG_MEL) * LOGSTEP);
}

extern "C" q3_status q3_spk_finalize(q3_speaker_encoder* e) {
    if (!e) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_finalize: null");
    if (e->device < 0) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_finalize: manifest-only handle");
    for (auto& s : e->slots)
        if (!s.loaded) return q3i_set_err(Q3_MISSING_WEIGHT, "Missing weight: %s", s.name.c_str());
    if (e->finalized) return Q3_OK;
    SPK_HIP(hipSetDevice(e->device));
    const q3_spk_config& c = e->cfg;
    std::vector<SpkConv*> all;
    e->c0 = spk_conv(e, "speaker_encoder.blocks.0.conv", c.channels[0], c.mel_dim, c.kernel_sizes[0], c.dilations[0]); all.push_back(&e->c0);
    for (int bi = 1; bi <= 3; ++bi) {
        SpkBlock& b = e->blk[bi - 1];
        const int C = c.channels[bi], chn = C / c.res2net_scale;
        b.tdnn1 = spk_conv(e, fmts("speaker_encoder.blocks.%d.tdnn1.conv", bi), C, C, 1, 1);
        b.res.clear();
        for (int i = 0; i < c.res2net_scale - 1; ++i)
            b.res.push_back(spk_conv(e, fmts("speaker_encoder.blocks.%d.res2net_block.blocks.%d.conv", bi, i), chn, chn, c.kernel_sizes[bi], c.dilations[bi]));
        b.tdnn2 = spk_conv(e, fmts("speaker_encoder.blocks.%d.tdnn2.conv", bi), C, C, 1, 1);
        b.se1 = spk_conv(e, fmts("speaker_encoder.blocks.%d.se_block.conv1", bi), c.se_channels, C, 1, 1);
        b.se2 = spk_conv(e, fmts("speaker_encoder.blocks.%d.se_block.conv2", bi), C, c.se_channels, 1, 1);
        all.push_back(&b.tdnn1); all.push_back(&b.tdnn2); all.push_back(&b.se1); all.push_back(&b.se2);
        for (auto& r : b.res) all.push_back(&r);
    }
    const int mfa_in = c.channels[1] + c.channels[2] + c.channels[3], C4 = c.channels[4];
    e->mfa = spk_conv(e, "speaker_encoder.mfa.conv", C4, mfa_in, c.kernel_sizes[4], c.dilations[4]); all.push_back(&e->mfa);
    e->asp_tdnn = spk_conv(e, "speaker_encoder.asp.tdnn.conv", c.attention_channels, 3 * C4, 1, 1); all.push_back(&e->asp_tdnn);
    e->asp_conv = spk_conv(e, "speaker_encoder.asp.conv", C4, c.attention_channels, 1, 1); all.push_back(&e->asp_conv);
    e->fc = spk_conv(e, "speaker_encoder.fc", c.enc_dim, 2 * C4, 1, 1); all.push_back(&e->fc);
    // bf16x3 matrix-core weight images for every conv the packed layout covers (the rest run the f32 kernels)
    size_t pk_bytes = 0;
    for (auto* cv : all)
        if (cv->cout % 32 == 0 && cv->cin % 16 == 0) pk_bytes += (packed_conv_w_bytes(cv->cout, cv->cin, cv->k) + 255) & ~(size_t)255;
    (void)hipFree(e->wpk_arena); e->wpk_arena = nullptr;
    if (pk_bytes) {
        SPK_HIP(hipMalloc(&e->wpk_arena, pk_bytes));
        size_t off = 0;
        for (auto* cv : all) {
            if (cv->cout % 32 || cv->cin % 16) continue;
            void* dst = (char*)e->wpk_arena + off;
            SPK_HIP(launch_pack_conv_w(cv->w, dst, cv->cout, cv->cin, cv->k, e->st));
            cv->wpk = dst;
            off += (packed_conv_w_bytes(cv->cout, cv->cin, cv->k) + 255) & ~(size_t)255;
        }
    }
    // front-end tables: periodic Hann (mel.rs:320-324, f32 arithmetic), DFT twiddles (f64), Slaney filterbank (mel.rs:271-318)
    constexpr int NF = 1024, NB = NF / 2 + 1;
    std::vector<float> win(NF), fb((size_t)c.mel_dim * NB, 0.0f);
    std::vector<double> cs(NF), sn(NF);
    const float PI_F = 3.14159265358979323846f;
    for (int i = 0; i < NF; ++i) {
        win[i] = 0.5f * (1.0f - cosf(2.0f * PI_F * (float)i / (float)NF));
        cs[i] = cos(2.0 * M_PI * i / NF); sn[i] = sin(2.0 * M_PI * i / NF);
    }
    {
        const int n_mels = c.mel_dim; const float sr = (float)c.sample_rate;
        const float mel_min = spk_hz_to_mel(0.0f), mel_max = spk_hz_to_mel(sr / 2.0f);
        std::vector<float> hz(n_mels + 2);
        for (int i = 0; i <= n_mels + 1; ++i) hz[i] = spk_mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(n_mels + 1));
        for (int i = 0; i < n_mels; ++i) {
            const float lo = hz[i], ce = hz[i + 1], up = hz[i + 2];
            float* row = fb.data() + (size_t)i * NB;
            for (int j = 0; j < NB; ++j) {
                const float freq = (float)j * sr / (float)NF;
                if (freq >= lo && freq <= ce && ce > lo) row[j] = (freq - lo) / (ce - lo);
                else if (freq > ce && freq <= up && up > ce) row[j] = (up - freq) / (up - ce);
            }
            const float bw = hz[i + 2] - hz[i];
            if (bw > 0.0f) { const float en = 2.0f / bw; for (int j = 0; j < NB; ++j) row[j] *= en; }
        }
    }
    if (!e->win) {
        SPK_HIP(hipMalloc((void**)&e->win, NF * 4)); SPK_HIP(hipMalloc((void**)&e->dft_cs, NF * 8)); SPK_HIP(hipMalloc((void**)&e->dft_sn, NF * 8));
        SPK_HIP(hipMalloc((void**)&e->fb, fb.size() * 4));
    }
    SPK_HIP(q3_hipMemcpy(e->win, win.data(), NF * 4, hipMemcpyHostToDevice));
    SPK_HIP(q3_hipMemcpy(e->dft_cs, cs.data(), NF * 8, hipMemcpyHostToDevice));
    SPK_HIP(q3_hipMemcpy(e->dft_sn, sn.data(), NF * 8, hipMemcpyHostToDevice));
    SPK_HIP(q3_hipMemcpy(e->fb, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
    SPK_HIP(hipStreamSynchronize(e->st));
    e->finalized = true;
    return Q3_OK;
}

extern "C" int q3_spk_mel_frames(int64_t n) {
    const int64_t padded = n + 2 * 384;
    return (n < 1 || padded < 1024) ? 0 : (int)((padded - 1024) / 256 + 1);
}

static q3_status spk_reserve(q3_speaker_encoder* e, size_t floats) {
    if (floats <= e->ws_floats) return Q3_OK;
    (void)hipFree(e->ws); e->ws = nullptr; e->ws_floats = 0;
    SPK_HIP(hipMalloc((void**)&e->ws, floats * 4));
    e->ws_floats = floats;
    return Q3_OK;
}

// ReflectPadConv1d::forward (speaker.rs:97-105) + the TDNN's ReLU: x [cin][T] (+ x2) -> y [cout][T]
static q3_status spk_same_conv(q3_speaker_encoder* e, const SpkConv& cv, const float* x, const float* x2, float* y, int T, int act,
                               float* pad_buf, float* tmp_buf) {
    ConvArgs a{};
    a.w = cv.w; a.b = cv.b; a.cin = cv.cin; a.cout = cv.cout; a.k = cv.k; a.dil = cv.dil; a.act = act; a.wpk = cv.wpk;
    const int tot = cv.dil * (cv.k - 1);
    if (tot == 0 && !x2) {
        a.x = x; a.y = y; a.L = T;
        SPK_HIP(launch_conv1d(a, e->st));
        return Q3_OK;
    }
    const int pl = tot / 2, Tp = T + tot;
    hipLaunchKernelGGL(k_spk_pad, dim3((Tp + 255) / 256, cv.cin), dim3(256), 0, e->st, x, x2, pad_buf, T, pl, Tp);
    a.x = pad_buf; a.L = Tp;
    if (tot == 0) {
        a.y = y;
        SPK_HIP(launch_conv1d(a, e->st));
        return Q3_OK;
    }
    // causal conv over the padded signal: column t + tot of its output is the "valid" conv at t
    a.y = tmp_buf;
    SPK_HIP(launch_conv1d(a, e->st));
    hipLaunchKernelGGL(k_spk_cols, dim3((T + 255) / 256, cv.cout), dim3(256), 0, e->st, tmp_buf, Tp, tot, y, T);
    SPK_HIP(hipGetLastError());
    return Q3_OK;
}

static q3_status spk_forward_dev(q3_speaker_encoder* e, const float* mel_d, int T, float* out_d, float* base, float** taps_host) {
    const q3_spk_config& c = e->cfg;
    const int C = c.channels[0], C4 = c.channels[4], mfa_in = c.channels[1] + c.channels[2] + c.channels[3];
    const int chn = C / c.res2net_scale, A = c.attention_channels;
    int maxtot = 0, max_cin_pad = c.mel_dim, max_cout_pad = C;
    for (int i = 0; i < 5; ++i) maxtot = std::max(maxtot, c.dilations[i] * (c.kernel_sizes[i] - 1));
    max_cin_pad = std::max(std::max(c.mel_dim, chn), c.kernel_sizes[4] > 1 ? mfa_in : 0);
    max_cout_pad = std::max(C, c.kernel_sizes[4] > 1 ? C4 : 0);
    const size_t Tp = (size_t)T + maxtot;
    float* p = base;
    auto take = [&](size_t n) { float* r = p; p += (n + 63) & ~(size_t)63; return r; };
    float* pad_buf = take((size_t)max_cin_pad * Tp);
    float* tmp_buf = take((size_t)max_cout_pad * Tp);
    float* h = take((size_t)C * T); float* o1 = take((size_t)C * T); float* o2 = take((size_t)C * T);
    float* cat = take((size_t)mfa_in * T); float* m = take((size_t)C4 * T); float* ain = take((size_t)3 * C4 * T);
    float* at = take((size_t)A * T); float* aw = take((size_t)C4 * T);
    float* sm = take(C); float* s1 = take(c.se_channels); float* g = take(C); float* pooled = take((size_t)2 * C4);
    auto tap = [&](int i, const float* src, size_t n) -> q3_status {
        if (taps_host && taps_host[i]) { SPK_HIP(hipMemcpyAsync(taps_host[i], src, n * 4, hipMemcpyDeviceToHost, e->st)); SPK_HIP(hipStreamSynchronize(e->st)); }
        return Q3_OK;
    };
    Q3I_CHECK(spk_same_conv(e, e->c0, mel_d, nullptr, h, T, 3, pad_buf, tmp_buf));                     // blocks.0: TDNN (speaker.rs:445)
    Q3I_CHECK(tap(0, h, (size_t)C * T));
    int cat_off = 0;
    for (int bi = 0; bi < 3; ++bi) {                                                                     // SE-Res2Net blocks (speaker.rs:262-272)
        const SpkBlock& b = e->blk[bi];
        Q3I_CHECK(spk_same_conv(e, b.tdnn1, h, nullptr, o1, T, 3, pad_buf, tmp_buf));
        SPK_HIP(hipMemcpyAsync(o2, o1, (size_t)chn * T * 4, hipMemcpyDeviceToDevice, e->st));             // first chunk passes through
        for (int i = 0; i < (int)b.res.size(); ++i) {
            const float* chunk = o1 + (size_t)(i + 1) * chn * T;
            const float* prev = i == 0 ? nullptr : o2 + (size_t)i * chn * T;
            Q3I_CHECK(spk_same_conv(e, b.res[i], chunk, prev, o2 + (size_t)(i + 1) * chn * T, T, 3, pad_buf, tmp_buf));
        }
        Q3I_CHECK(spk_same_conv(e, b.tdnn2, o2, nullptr, o1, T, 3, pad_buf, tmp_buf));
        hipLaunchKernelGGL(k_spk_mean, dim3(C), dim3(256), 0, e->st, o1, sm, T);                         // SE (speaker.rs:218-226)
        Q3I_CHECK(spk_same_conv(e, b.se1, sm, nullptr, s1, 1, 3, pad_buf, tmp_buf));
        Q3I_CHECK(spk_same_conv(e, b.se2, s1, nullptr, g, 1, 5, pad_buf, tmp_buf));
        hipLaunchKernelGGL(k_spk_se_apply, dim3((T + 255) / 256, C), dim3(256), 0, e->st, o1, g, h, T);
        SPK_HIP(hipMemcpyAsync(cat + (size_t)cat_off * T, h, (size_t)C * T * 4, hipMemcpyDeviceToDevice, e->st));
        cat_off += C;
        Q3I_CHECK(tap(1 + bi, h, (size_t)C * T));
    }
    Q3I_CHECK(spk_same_conv(e, e->mfa, cat, nullptr, m, T, 3, pad_buf, tmp_buf));                       // MFA (speaker.rs:455-459)
    Q3I_CHECK(tap(4, m, (size_t)C4 * T));
    hipLaunchKernelGGL(k_spk_asp_in, dim3(C4), dim3(256), 0, e->st, m, ain, C4, T);                       // ASP (speaker.rs:301-343)
    Q3I_CHECK(spk_same_conv(e, e->asp_tdnn, ain, nullptr, at, T, 4, pad_buf, tmp_buf));                  // ReLU then tanh
    Q3I_CHECK(spk_same_conv(e, e->asp_conv, at, nullptr, aw, T, 0, pad_buf, tmp_buf));
    hipLaunchKernelGGL(k_spk_asp_pool, dim3(C4), dim3(256), 0, e->st, aw, m, pooled, C4, T);
    Q3I_CHECK(tap(5, pooled, (size_t)2 * C4));
    Q3I_CHECK(spk_same_conv(e, e->fc, pooled, nullptr, out_d, 1, 0, pad_buf, tmp_buf));                 // FC (speaker.rs:464-465)
    SPK_HIP(hipGetLastError());
    return Q3_OK;
}

static size_t spk_ws_floats(const q3_speaker_encoder* e, int T) {
    const q3_spk_config& c = e->cfg;
    const int C = c.channels[0], C4 = c.channels[4], mfa_in = c.channels[1] + c.channels[2] + c.channels[3];
    int maxtot = 0;
    for (int i = 0; i < 5; ++i) maxtot = std::max(maxtot, c.dilations[i] * (c.kernel_sizes[i] - 1));
    const size_t Tp = (size_t)T + maxtot;
    size_t n = (size_t)std::max(std::max(c.mel_dim, C), mfa_in) * Tp + (size_t)std::max(C, C4) * Tp;
    n += (size_t)3 * C * T + (size_t)mfa_in * T + (size_t)C4 * T * 2 + (size_t)3 * C4 * T + (size_t)c.attention_channels * T;
    n += (size_t)2 * C + c.se_channels + 2 * C4 + c.enc_dim + (size_t)c.mel_dim * T;
    return n + 64 * 24;
}

static q3_status spk_check_T(const q3_speaker_encoder* e, int T) {
    int maxside = 0;
    for (int i = 0; i < 5; ++i) maxside = std::max(maxside, (e->cfg.dilations[i] * (e->cfg.kernel_sizes[i] - 1) + 1) / 2);
    if (T <= maxside) return q3i_set_err(Q3_INVALID_ARG, "reference audio too short: %d mel frames, reflect padding needs more than %d", T, maxside);
    return Q3_OK;
}

extern "C" q3_status q3_spk_mel(q3_speaker_encoder* e, const float* samples, int64_t n, float* mel_host, int64_t cap, int* n_frames) {
    if (!e || !samples || n < 1) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_mel: bad argument");
    if (!e->finalized) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_mel: encoder not finalized");
    const int T = q3_spk_mel_frames(n);
    if (n_frames) *n_frames = T;
    if (!mel_host) return Q3_OK;
    if (T < 1 || cap < (int64_t)e->cfg.mel_dim * T) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_mel: output buffer too small (%d frames)", T);
    SPK_HIP(hipSetDevice(e->device));
    Q3I_CHECK(spk_reserve(e, (size_t)n + 64 + (size_t)e->cfg.mel_dim * T));
    float* x_d = e->ws; float* mel_d = e->ws + (((size_t)n + 63) & ~(size_t)63);
    SPK_HIP(hipMemcpyAsync(x_d, samples, (size_t)n * 4, hipMemcpyHostToDevice, e->st));
    hipLaunchKernelGGL(k_spk_stft_mel, dim3(T), dim3(256), 0, e->st, x_d, (long)n, T, e->win, e->dft_cs, e->dft_sn, e->fb, mel_d, e->cfg.mel_dim);
    SPK_HIP(hipGetLastError());
    SPK_HIP(hipMemcpyAsync(mel_host, mel_d, (size_t)e->cfg.mel_dim * T * 4, hipMemcpyDeviceToHost, e->st));
    SPK_HIP(hipStreamSynchronize(e->st));
    return Q3_OK;
}

extern "C" q3_status q3_spk_forward(q3_speaker_encoder* e, const float* mel_host, int T, float* out_host, float** taps_host) {
    if (!e || !mel_host || !out_host || T < 1) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_forward: bad argument");
    if (!e->finalized) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_forward: encoder not finalized");
    Q3I_CHECK(spk_check_T(e, T));
    SPK_HIP(hipSetDevice(e->device));
    Q3I_CHECK(spk_reserve(e, spk_ws_floats(e, T)));
    float* mel_d = e->ws; float* out_d = mel_d + (((size_t)e->cfg.mel_dim * T + 63) & ~(size_t)63);
    float* base = out_d + ((e->cfg.enc_dim + 63) & ~63);
    SPK_HIP(hipMemcpyAsync(mel_d, mel_host, (size_t)e->cfg.mel_dim * T * 4, hipMemcpyHostToDevice, e->st));
    Q3I_CHECK(spk_forward_dev(e, mel_d, T, out_d, base, taps_host));
    SPK_HIP(hipMemcpyAsync(out_host, out_d, (size_t)e->cfg.enc_dim * 4, hipMemcpyDeviceToHost, e->st));
    SPK_HIP(hipStreamSynchronize(e->st));
    return Q3_OK;
}

extern "C" q3_status q3_spk_encode(q3_speaker_encoder* e, const float* samples, int64_t n, uint32_t sample_rate, float* out_host) {
    if (!e || !samples || !out_host || n < 1) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_encode: bad argument");
    if (!e->finalized) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_encode: encoder not finalized");
    if (sample_rate != (uint32_t)e->cfg.sample_rate)
        return q3i_set_err(Q3_UNSUPPORTED, "speaker encoder expects %d Hz audio, got %u Hz: resample first (the reference does, lib.rs:1156-1166)", e->cfg.sample_rate, sample_rate);
    const int T = q3_spk_mel_frames(n);
    Q3I_CHECK(spk_check_T(e, T));
    SPK_HIP(hipSetDevice(e->device));
    const size_t xs = ((size_t)n + 63) & ~(size_t)63;
    Q3I_CHECK(spk_reserve(e, xs + spk_ws_floats(e, T)));
    float* x_d = e->ws; float* mel_d = x_d + xs; float* out_d = mel_d + (((size_t)e->cfg.mel_dim * T + 63) & ~(size_t)63);
    float* base = out_d + ((e->cfg.enc_dim + 63) & ~63);
    SPK_HIP(hipMemcpyAsync(x_d, samples, (size_t)n * 4, hipMemcpyHostToDevice, e->st));
    hipLaunchKernelGGL(k_spk_stft_mel, dim3(T), dim3(256), 0, e->st, x_d, (long)n, T, e->win, e->dft_cs, e->dft_sn, e->fb, mel_d, e->cfg.mel_dim);
    Q3I_CHECK(spk_forward_dev(e, mel_d, T, out_d, base, nullptr));
    SPK_HIP(hipMemcpyAsync(out_host, out_d, (size_t)e->cfg.enc_dim * 4, hipMemcpyDeviceToHost, e->st));
    SPK_HIP(hipStreamSynchronize(e->st));
    return Q3_OK;
}
