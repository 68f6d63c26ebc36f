// q3_mimi.hip — the speech-tokenizer ENCODER on gfx950: 24 kHz reference audio -> 12.5 Hz frames of 16 codebook indices,
// the `ref_codes` of an ICL voice-clone prompt (reference: Encoder12Hz, src/models/codec/encoder_12hz.rs:34-144, called by
// create_voice_clone_prompt, src/lib.rs:1172-1178). The reference builds it from candle-transformers' Mimi modules over the
// HF-format `encoder.*` keys of speech_tokenizer/model.safetensors; the algorithm is the published Mimi encoder (Hugging
// Face transformers models/mimi/modeling_mimi.py): SEANet conv encoder -> 8-layer transformer -> stride-2 conv -> split
// residual vector quantiser (nearest neighbour per layer). oracle/q3_oracle_mimi.c restates it for the parity tests.
//
// Once-per-utterance work (~16 GFLOP for 5 s of audio), so it rides the vocoder's kernels instead of growing its own GEMMs:
//   * stride-1 causal convs  -> launch_conv1d (bf16x3 matrix-core implicit GEMM where the channel counts allow);
//   * strided convs (kernel 2r, stride r) -> the input is FOLDED r samples into channels (k_mimi_fold, with the ELU that
//     precedes every such conv applied on the way), which turns the conv into a 2-tap stride-1 causal conv with cin*r
//     input channels — the weights are re-laid once at upload: w'[co][ci*r + p][j] = w[co][ci][j*r + p];
//   * transformer layers on the [C][T] layout as 1x1 convs + channel LayerNorm + rotate-half RoPE, attention with the
//     250-frame causal window in k_mimi_attn;
//   * k_mimi_rvq: one workgroup per frame walks the quantiser layers (distance to 2048 entries, first minimum, subtract).
#include "../../include/q3tts.h"
#include "q3_kernels.h"
#include "q3_capture_lock.h"
#include "q3_internal.h"

#include <math.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

namespace q3 {

// y = ELU(x) (alpha 1)
__global__ __launch_bounds__(256) void k_mimi_elu(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v > 0.0f ? v : expf(v) - 1.0f; }
}
// fold r consecutive samples into channels: y[(c*r + p)][lead + tau] = f(x[c][tau*r + p]) (0 beyond L), f = ELU or identity.
// lead = 1 (replicate mode): column 0 holds x[c][0] for every phase — the replicate left padding of the downsample conv —
// and samples beyond L repeat x[c][L-1].
__global__ __launch_bounds__(256) void k_mimi_fold(const float* __restrict__ x, float* __restrict__ y, int L, int r, int Lf, int elu, int replicate) {
    const int col = blockIdx.x * 256 + threadIdx.x, cp = blockIdx.y;     // cp = c*r + p
    const int lead = replicate ? 1 : 0, W = Lf + lead;
    if (col >= W) return;
    const int c = cp / r, p = cp % r;
    const float* xr = x + (size_t)c * L;
    float v;
    if (replicate && col == 0) v = xr[0];
    else {
        const int i = (col - lead) * r + p;
        v = i < L ? xr[i] : (replicate ? xr[L - 1] : 0.0f);
    }
    if (elu) v = v > 0.0f ? v : expf(v) - 1.0f;
    y[(size_t)cp * W + col] = v;
}
// causal sliding-window MHA on [nh*64][T] tensors: one 64-lane wave per (query t, head h), keys j in (t - window, t]
__global__ __launch_bounds__(64) void k_mimi_attn(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                  float* __restrict__ o, int T, int window, float scale) {
    const int t = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const size_t hb = (size_t)h * 64 * T;
    const float qd = q[hb + (size_t)lane * T + t];
    int j0 = t - window + 1; if (j0 < 0) j0 = 0;
    float m = -INFINITY, l = 0.0f, acc = 0.0f;
    for (int jb = j0; jb <= t; jb += 64) {
        const int j = jb + lane; const bool ok = j <= t;
        float s = 0.0f;
        for (int d = 0; d < 64; ++d) s = fmaf(__shfl(qd, d), ok ? k[hb + (size_t)d * T + j] : 0.0f, s);
        s = ok ? s * scale : -INFINITY;
        float cm = s;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cm = fmaxf(cm, __shfl_xor(cm, off));
        const float mn = fmaxf(m, cm), corr = expf(m - mn), p = ok ? expf(s - mn) : 0.0f;
        float ps = p;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ps += __shfl_xor(ps, off);
        l = l * corr + ps; acc *= corr;
        for (int d = 0; d < 64; ++d) {
            float c = ok ? p * v[hb + (size_t)d * T + j] : 0.0f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
            if (lane == d) acc += c;
        }
        m = mn;
    }
    o[hb + (size_t)lane * T + t] = acc / l;
}
// codebook = embed_sum / max(cluster_usage, 1e-5)  (MimiEuclideanCodebook.embed)
__global__ __launch_bounds__(256) void k_mimi_codebook(const float* __restrict__ esum, const float* __restrict__ usage, float* __restrict__ out, int dim) {
    const int e = blockIdx.x;
    const float u = fmaxf(usage[e], 1e-5f);
    for (int d = threadIdx.x; d < dim; d += 256) out[(size_t)e * dim + d] = esum[(size_t)e * dim + d] / u;
}
// MimiResidualVectorQuantizer.encode for one group: proj [CD][T] (input_proj output), books = n_layers tables [CB][CD];
// codes[t*n_q + q0 + l] = argmin_e |r - e|^2 (first minimum), r -= e. One workgroup per frame.
__global__ __launch_bounds__(256) void k_mimi_rvq(const float* __restrict__ proj, int T, const float* __restrict__ books, int n_layers, int CB, int CD,
                                                  uint32_t* __restrict__ codes, int n_q, int q0) {
    __shared__ float r[256];
    __shared__ float bd[256]; __shared__ int bi[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (tid < CD) r[tid] = proj[(size_t)tid * T + t];
    __syncthreads();
    for (int l = 0; l < n_layers; ++l) {
        const float* book = books + (size_t)l * CB * CD;
        float best = INFINITY; int bidx = 0x7fffffff;
        for (int e = tid; e < CB; e += 256) {
            const float* ev = book + (size_t)e * CD;
            float d2 = 0.0f;
            for (int d = 0; d < CD; ++d) { const float df = sub_rn(r[d], ev[d]); d2 = add_rn(d2, mul_rn(df, df)); }
            if (d2 < best) { best = d2; bidx = e; }            // ascending e per thread: the first minimum stays
        }
        bd[tid] = best; bi[tid] = bidx;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (tid < s) {
                const float ob = bd[tid + s]; const int oi = bi[tid + s];
                if (ob < bd[tid] || (ob == bd[tid] && oi < bi[tid])) { bd[tid] = ob; bi[tid] = oi; }
            }
            __syncthreads();
        }
        const int win = bi[0];
        if (tid == 0) codes[(size_t)t * n_q + q0 + l] = (uint32_t)win;
        if (tid < CD) r[tid] = sub_rn(r[tid], book[(size_t)win * CD + tid]);
        __syncthreads();
    }
}

}  // namespace q3

using namespace q3;

namespace {
struct MSlot { std::string name; int64_t n = 0; size_t offset = 0; bool loaded = false; int fold_r = 0, cout = 0, cin = 0, k = 0; };
struct MConv { const float* w = nullptr; const float* b = nullptr; const void* wpk = nullptr; int cin = 0, cout = 0, k = 1; };
struct MLayer { const float *ln1w, *ln1b, *ln2w, *ln2b, *sa, *sm; MConv q, k, v, o, f1, f2; };
std::string fm(const char* f, int a, int b = 0) { char buf[192]; snprintf(buf, sizeof buf, f, a, b); return buf; }
}  // namespace

struct q3_speech_encoder {
    q3_mimi_config cfg{};
    int device = 0;
    std::vector<MSlot> slots;
    std::unordered_map<std::string, int> index;
    char* arena = nullptr; size_t arena_bytes = 0;
    void* wpk_arena = nullptr;
    float* books[2] = {nullptr, nullptr};           // normalised codebooks per group
    bool finalized = false;
    hipStream_t st = nullptr;
    MConv c0, res1[4], res2[4], down[4], last, dsamp, qproj[2];
    std::vector<MLayer> layers;
    float* ws = nullptr; size_t ws_floats = 0;
    float *cs = nullptr, *sn = nullptr; int rope_T = 0;
    uint32_t* codes_dev = nullptr; int codes_cap = 0;
};

#define M_HIP(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return q3i_set_err(Q3_HIP_ERROR, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static void m_add(q3_speech_encoder* e, const std::string& name, int64_t n, int fold_r = 0, int cout = 0, int cin = 0, int k = 0) {
    MSlot s; s.name = name; s.n = n; s.fold_r = fold_r; s.cout = cout; s.cin = cin; s.k = k;
    s.offset = (e->arena_bytes + 255) & ~(size_t)255;
    e->arena_bytes = s.offset + (size_t)n * 4;
    e->index[name] = (int)e->slots.size();
    e->slots.push_back(s);
}

extern "C" q3_status q3_mimi_config_default(q3_mimi_config* out) {
    if (!out) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_config_default: null");
    q3_mimi_config c{};
    c.n_filters = 64; c.hidden = 512; c.ratios[0] = 4; c.ratios[1] = 5; c.ratios[2] = 6; c.ratios[3] = 8;
    c.kernel = 7; c.res_kernel = 3; c.last_kernel = 3; c.compress = 2;
    c.n_layers = 8; c.n_heads = 8; c.head_dim = 64; c.inter = 2048; c.window = 250;
    c.cb_size = 2048; c.cb_dim = 256; c.n_q = 16; c.n_sem = 1; c.norm_eps = 1e-5f; c.rope_theta = 1e4f;
    *out = c;
    return Q3_OK;
}

extern "C" void q3_mimi_free(q3_speech_encoder* e) {
    if (!e) return;
    if (e->device >= 0) {
        (void)hipSetDevice(e->device);
        if (e->st) { (void)hipStreamSynchronize(e->st); (void)hipStreamDestroy(e->st); }
        (void)hipFree(e->arena); (void)hipFree(e->wpk_arena); (void)hipFree(e->books[0]); (void)hipFree(e->books[1]);
        (void)hipFree(e->ws); (void)hipFree(e->cs); (void)hipFree(e->sn); (void)hipFree(e->codes_dev);
    }
    delete e;
}

extern "C" q3_status q3_mimi_create(const q3_mimi_config* cfg, int device, q3_speech_encoder** out) {
    if (!cfg || !out) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_create: null argument");
    const q3_mimi_config& c = *cfg;
    if (c.n_filters < 1 || c.hidden < 1 || c.compress < 1 || c.n_filters % c.compress || c.kernel < 1 || c.res_kernel < 1 || c.last_kernel < 1 ||
        c.n_layers < 0 || c.n_heads < 1 || c.inter < 1 || c.window < 1 || c.cb_size < 2 || c.cb_dim < 1 || c.cb_dim > 256 || c.n_q < 1 ||
        c.n_sem < 1 || c.n_sem > c.n_q || c.n_q > 64)
        return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_create: bad config");
    for (int i = 0; i < 4; ++i) if (c.ratios[i] < 1 || c.ratios[i] > 16) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_create: bad ratio");
    if (c.head_dim != 64) return q3i_set_err(Q3_UNSUPPORTED, "speech encoder head_dim %d unsupported (64)", c.head_dim);
    if (c.n_heads * c.head_dim != c.hidden) return q3i_set_err(Q3_UNSUPPORTED, "speech encoder: n_heads * head_dim must equal hidden");
    auto* e = new q3_speech_encoder();
    e->cfg = c; e->device = device;
    const int F = c.n_filters, H = c.hidden;
    auto conv = [&](const std::string& p, int cout, int cin, int k, bool bias, int fold_r = 0) {
        m_add(e, p + ".weight", (int64_t)cout * cin * k, fold_r, cout, cin, k);
        if (bias) m_add(e, p + ".bias", cout);
    };
    conv("encoder.encoder.layers.0.conv", F, 1, c.kernel, true);
    int dim = F, li = 1;
    for (int s = 0; s < 4; ++s) {
        conv(fm("encoder.encoder.layers.%d.block.1.conv", li), dim / c.compress, dim, c.res_kernel, true);
        conv(fm("encoder.encoder.layers.%d.block.3.conv", li), dim, dim / c.compress, 1, true);
        conv(fm("encoder.encoder.layers.%d.conv", li + 2), 2 * dim, dim, 2 * c.ratios[s], true, c.ratios[s]);
        dim *= 2; li += 3;
    }
    conv(fm("encoder.encoder.layers.%d.conv", li + 1), H, dim, c.last_kernel, true);
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string p = fm("encoder.encoder_transformer.layers.%d", l);
        conv(p + ".self_attn.q_proj", H, H, 1, false); conv(p + ".self_attn.k_proj", H, H, 1, false);
        conv(p + ".self_attn.v_proj", H, H, 1, false); conv(p + ".self_attn.o_proj", H, H, 1, false);
        conv(p + ".mlp.fc1", c.inter, H, 1, false); conv(p + ".mlp.fc2", H, c.inter, 1, false);
        m_add(e, p + ".input_layernorm.weight", H); m_add(e, p + ".input_layernorm.bias", H);
        m_add(e, p + ".post_attention_layernorm.weight", H); m_add(e, p + ".post_attention_layernorm.bias", H);
        m_add(e, p + ".self_attn_layer_scale.scale", H); m_add(e, p + ".mlp_layer_scale.scale", H);
    }
    conv("encoder.downsample.conv", H, H, 4, false, 2);
    for (int g = 0; g < 2; ++g) {
        const std::string p = std::string("encoder.quantizer.") + (g ? "acoustic" : "semantic") + "_residual_vector_quantizer";
        conv(p + ".input_proj", c.cb_dim, H, 1, false);
        const int nl = g ? c.n_q - c.n_sem : c.n_sem;
        for (int l = 0; l < nl; ++l) {
            m_add(e, p + fm(".layers.%d.codebook.embed_sum", l), (int64_t)c.cb_size * c.cb_dim);
            m_add(e, p + fm(".layers.%d.codebook.cluster_usage", l), c.cb_size);
        }
    }
    if (device >= 0) {
        hipError_t he = hipSetDevice(device);
        if (he == hipSuccess) he = hipMalloc((void**)&e->arena, e->arena_bytes);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking);
        if (he != hipSuccess) { q3_mimi_free(e); return q3i_set_err(Q3_HIP_ERROR, "q3_mimi_create: %s", hipGetErrorString(he)); }
    }
    *out = e;
    return Q3_OK;
}

extern "C" q3_status q3_mimi_get_config(const q3_speech_encoder* e, q3_mimi_config* out) {
    if (!e || !out) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_get_config: null");
    *out = e->cfg;
    return Q3_OK;
}
extern "C" int q3_mimi_n_tensors(const q3_speech_encoder* e) { return e ? (int)e->slots.size() : 0; }
extern "C" q3_status q3_mimi_tensor_info(const q3_speech_encoder* e, int i, const char** name, int64_t* n) {
    if (!e || i < 0 || i >= (int)e->slots.size()) return q3i_set_err(Q3_INVALID_ARG, "speech-encoder tensor index out of range");
    if (name) *name = e->slots[i].name.c_str();
    if (n) *n = e->slots[i].n;
    return Q3_OK;
}

extern "C" q3_status q3_mimi_set_tensor(q3_speech_encoder* e, const char* name, const void* data, int src_dtype, int64_t n) {
    if (!e || !name || !data) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_set_tensor: null argument");
    if (e->device < 0) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_set_tensor: manifest-only handle");
    auto it = e->index.find(name);
    if (it == e->index.end()) return q3i_set_err(Q3_INVALID_ARG, "unknown speech-encoder tensor %s", name);
    MSlot& s = e->slots[it->second];
    if (s.n != n) return q3i_set_err(Q3_INVALID_ARG, "tensor %s: expected %lld elements, got %lld", name, (long long)s.n, (long long)n);
    M_HIP(hipSetDevice(e->device));
    std::vector<float> tmp((size_t)n);
    if (src_dtype == Q3_DTYPE_F32) memcpy(tmp.data(), data, (size_t)n * 4);
    else if (src_dtype == Q3_DTYPE_BF16) {
        const uint16_t* h = (const uint16_t*)data;
        for (int64_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)h[i] << 16; memcpy(&tmp[(size_t)i], &u, 4); }
    } else return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_set_tensor: unsupported source dtype %d", src_dtype);
    if (s.fold_r > 0) {
        // strided conv [cout][cin][2r] -> folded 2-tap causal conv [cout][cin*r][2]: w'[co][ci*r + p][j] = w[co][ci][j*r + p]
        const int r = s.fold_r, cout = s.cout, cin = s.cin;
        std::vector<float> f((size_t)n);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int j = 0; j < 2; ++j)
                    for (int p = 0; p < r; ++p)
                        f[(((size_t)co * cin + ci) * r + p) * 2 + j] = tmp[((size_t)co * cin + ci) * 2 * r + (size_t)j * r + p];
        tmp.swap(f);
    }
    M_HIP(q3_hipMemcpy(e->arena + s.offset, tmp.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    s.loaded = true; e->finalized = false;
    return Q3_OK;
}

static MConv m_conv(q3_speech_encoder* e, const std::string& prefix, int cout, int cin, int k, bool bias) {
    MConv c; c.cout = cout; c.cin = cin; c.k = k;
    c.w = (const float*)(e->arena + e->slots[e->index[prefix + ".weight"]].offset);
    c.b = bias ? (const float*)(e->arena + e->slots[e->index[prefix + ".bias"]].offset) : nullptr;
    return c;
}
static const float* m_vec(q3_speech_encoder* e, const std::string& name) { return (const float*)(e->arena + e->slots[e->index[name]].offset); }

extern "C" q3_status q3_mimi_finalize(q3_speech_encoder* e) {
    if (!e) return q3i_set_err(Q3_INVALID_ARG, "null speech encoder");
    if (e->device < 0) return q3i_set_err(Q3_UNSUPPORTED, "manifest-only handle cannot be finalized");
    for (auto& s : e->slots) if (!s.loaded) return q3i_set_err(Q3_MISSING_WEIGHT, "Missing weight: %s", s.name.c_str());
    M_HIP(hipSetDevice(e->device));
    const q3_mimi_config& c = e->cfg;
    const int F = c.n_filters, H = c.hidden;
    e->c0 = m_conv(e, "encoder.encoder.layers.0.conv", F, 1, c.kernel, true);
    int dim = F, li = 1;
    for (int s = 0; s < 4; ++s) {
        e->res1[s] = m_conv(e, fm("encoder.encoder.layers.%d.block.1.conv", li), dim / c.compress, dim, c.res_kernel, true);
        e->res2[s] = m_conv(e, fm("encoder.encoder.layers.%d.block.3.conv", li), dim, dim / c.compress, 1, true);
        e->down[s] = m_conv(e, fm("encoder.encoder.layers.%d.conv", li + 2), 2 * dim, dim * c.ratios[s], 2, true);      // folded form
        dim *= 2; li += 3;
    }
    e->last = m_conv(e, fm("encoder.encoder.layers.%d.conv", li + 1), H, dim, c.last_kernel, true);
    e->layers.resize(c.n_layers);
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string p = fm("encoder.encoder_transformer.layers.%d", l);
        MLayer& L = e->layers[l];
        L.q = m_conv(e, p + ".self_attn.q_proj", H, H, 1, false); L.k = m_conv(e, p + ".self_attn.k_proj", H, H, 1, false);
        L.v = m_conv(e, p + ".self_attn.v_proj", H, H, 1, false); L.o = m_conv(e, p + ".self_attn.o_proj", H, H, 1, false);
        L.f1 = m_conv(e, p + ".mlp.fc1", c.inter, H, 1, false); L.f2 = m_conv(e, p + ".mlp.fc2", H, c.inter, 1, false);
        L.ln1w = m_vec(e, p + ".input_layernorm.weight"); L.ln1b = m_vec(e, p + ".input_layernorm.bias");
        L.ln2w = m_vec(e, p + ".post_attention_layernorm.weight"); L.ln2b = m_vec(e, p + ".post_attention_layernorm.bias");
        L.sa = m_vec(e, p + ".self_attn_layer_scale.scale"); L.sm = m_vec(e, p + ".mlp_layer_scale.scale");
    }
    e->dsamp = m_conv(e, "encoder.downsample.conv", H, H * 2, 2, false);        // folded form (stride 2)
    for (int g = 0; g < 2; ++g) {
        const std::string p = std::string("encoder.quantizer.") + (g ? "acoustic" : "semantic") + "_residual_vector_quantizer";
        e->qproj[g] = m_conv(e, p + ".input_proj", c.cb_dim, H, 1, false);
        const int nl = g ? c.n_q - c.n_sem : c.n_sem;
        if (!e->books[g] && nl > 0) M_HIP(hipMalloc((void**)&e->books[g], (size_t)nl * c.cb_size * c.cb_dim * 4));
        for (int l = 0; l < nl; ++l)
            hipLaunchKernelGGL(k_mimi_codebook, dim3(c.cb_size), dim3(256), 0, e->st, m_vec(e, p + fm(".layers.%d.codebook.embed_sum", l)),
                               m_vec(e, p + fm(".layers.%d.codebook.cluster_usage", l)), e->books[g] + (size_t)l * c.cb_size * c.cb_dim, c.cb_dim);
    }
    // bf16x3 matrix-core images of every conv whose channel counts allow it (cout % 32 == 0, cin % 16 == 0)
    std::vector<MConv*> all = {&e->c0, &e->last, &e->dsamp, &e->qproj[0], &e->qproj[1]};
    for (int s = 0; s < 4; ++s) { all.push_back(&e->res1[s]); all.push_back(&e->res2[s]); all.push_back(&e->down[s]); }
    for (auto& L : e->layers) { all.push_back(&L.q); all.push_back(&L.k); all.push_back(&L.v); all.push_back(&L.o); all.push_back(&L.f1); all.push_back(&L.f2); }
    size_t total = 0;
    for (MConv* cv : all) if (cv->cout % 32 == 0 && cv->cin % 16 == 0) total += packed_conv_w_bytes(cv->cout, cv->cin, cv->k);
    if (total) {
        if (e->wpk_arena) { M_HIP(hipFree(e->wpk_arena)); e->wpk_arena = nullptr; }
        M_HIP(hipMalloc(&e->wpk_arena, total));
        char* cur = (char*)e->wpk_arena;
        for (MConv* cv : all) {
            cv->wpk = nullptr;
            if (cv->cout % 32 || cv->cin % 16) continue;
            M_HIP(launch_pack_conv_w(cv->w, cur, cv->cout, cv->cin, cv->k, e->st));
            cv->wpk = cur; cur += packed_conv_w_bytes(cv->cout, cv->cin, cv->k);
        }
    }
    M_HIP(hipGetLastError());
    M_HIP(hipStreamSynchronize(e->st));
    e->finalized = true;
    return Q3_OK;
}

static int m_ceil_div(int64_t a, int b) { return (int)((a + b - 1) / b); }
extern "C" int q3_mimi_frames(const q3_mimi_config* c, int64_t n_samples) {
    if (!c || n_samples < 1) return 0;
    int64_t L = n_samples;
    for (int i = 0; i < 4; ++i) L = (L + c->ratios[i] - 1) / c->ratios[i];
    return (int)((L + 1) / 2);
}

static hipError_t m_run_conv(q3_speech_encoder* e, const MConv& cv, const float* x, float* y, int L, int act = 0, const float* resid = nullptr,
                             const float* scale = nullptr) {
    ConvArgs a{}; a.x = x; a.w = cv.w; a.b = cv.b; a.y = y; a.cin = cv.cin; a.cout = cv.cout; a.L = L; a.k = cv.k; a.dil = 1;
    a.act = act; a.resid = resid; a.scale = scale; a.wpk = cv.wpk;
    return launch_conv1d(a, e->st);
}

// codes_host [T][n_q]; taps_host: NULL or 3 host pointers (NULL entries skipped): SEANet out [hidden][T25], transformer out [hidden][T25],
// downsampled [hidden][T]
extern "C" q3_status q3_mimi_encode(q3_speech_encoder* e, const float* samples, int64_t n, uint32_t sample_rate, uint32_t* codes_host,
                                    int cap_frames, int* n_frames, float** taps_host) {
    if (!e || !samples || !n_frames) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_encode: null argument");
    if (!e->finalized) return q3i_set_err(Q3_INVALID_ARG, "speech encoder not finalized");
    if (sample_rate != 24000) return q3i_set_err(Q3_INVALID_ARG, "speech encoder expects 24000 Hz audio, got %u (resample first: q3_resample)", sample_rate);
    if (n < 1 || n > (int64_t)24000 * 120) return q3i_set_err(Q3_INVALID_ARG, "reference audio of %lld samples unsupported (1 .. 120 s)", (long long)n);
    const q3_mimi_config& c = e->cfg;
    const int T = q3_mimi_frames(&c, n);
    *n_frames = T;
    if (!codes_host) return Q3_OK;
    if (cap_frames < T) return q3i_set_err(Q3_INVALID_ARG, "codes buffer too small (%d < %d frames)", cap_frames, T);
    M_HIP(hipSetDevice(e->device));
    const int F = c.n_filters, H = c.hidden;
    // workspace: three ping-pong buffers of the largest [C][L] activation (+ folding slack), the transformer's q|k|v|att, the MLP hidden
    int Ls[6]; Ls[0] = (int)n;
    for (int s = 0; s < 4; ++s) Ls[s + 1] = m_ceil_div(Ls[s], c.ratios[s]);
    const int T25 = Ls[4];
    size_t big = 0; { int dim = F; for (int s = 0; s < 4; ++s) { const size_t v = (size_t)dim * ((size_t)Ls[s] + 16 * c.ratios[s]); if (v > big) big = v; dim *= 2; }
                      const size_t v = (size_t)dim * (Ls[4] + 2); if (v > big) big = v; }
    const size_t tf = (size_t)T25 + 4;
    const size_t need = 3 * big + (size_t)(5 * H + c.inter) * tf + (size_t)2 * H * tf + (size_t)c.cb_dim * (T + 2) + 64;
    if (need > e->ws_floats) {
        M_HIP(hipStreamSynchronize(e->st));
        if (e->ws) M_HIP(hipFree(e->ws));
        e->ws = nullptr; e->ws_floats = 0;
        M_HIP(hipMalloc((void**)&e->ws, need * 4));
        e->ws_floats = need;
    }
    if (T25 > e->rope_T) {
        if (e->cs) { M_HIP(hipFree(e->cs)); M_HIP(hipFree(e->sn)); e->cs = e->sn = nullptr; }
        const int cap = T25 + 64;
        std::vector<float> cs((size_t)cap * 32), sn((size_t)cap * 32);
        for (int i = 0; i < 32; ++i) {         // MimiRotaryEmbedding: inv_freq = theta^(-2i/64), freqs = pos * inv_freq in f32 (host libm, as the oracle)
            const float inv = 1.0f / powf(c.rope_theta, (float)(2 * i) / 64.0f);
            for (int t = 0; t < cap; ++t) { const float f = (float)t * inv; cs[(size_t)t * 32 + i] = cosf(f); sn[(size_t)t * 32 + i] = sinf(f); }
        }
        M_HIP(hipMalloc((void**)&e->cs, cs.size() * 4)); M_HIP(hipMalloc((void**)&e->sn, sn.size() * 4));
        M_HIP(q3_hipMemcpy(e->cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice)); M_HIP(q3_hipMemcpy(e->sn, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
        e->rope_T = cap;
    }
    if (T > e->codes_cap) {
        if (e->codes_dev) M_HIP(hipFree(e->codes_dev));
        e->codes_dev = nullptr; e->codes_cap = 0;
        M_HIP(hipMalloc((void**)&e->codes_dev, (size_t)(T + 64) * c.n_q * 4));
        e->codes_cap = T + 64;
    }
    float* A = e->ws; float* B = A + big; float* C = B + big; float* tr = C + big;
    hipStream_t st = e->st;
    auto TAP = [&](int id, const float* dev, size_t cnt) -> q3_status {
        if (taps_host && taps_host[id]) { M_HIP(hipStreamSynchronize(st)); M_HIP(q3_hipMemcpy(taps_host[id], dev, cnt * 4, hipMemcpyDeviceToHost)); }
        return Q3_OK;
    };
    // ---- SEANet ----
    M_HIP(hipMemcpyAsync(C, samples, (size_t)n * 4, hipMemcpyHostToDevice, st));
    M_HIP(m_run_conv(e, e->c0, C, A, (int)n));                                          // x = A [F][n]
    float* x = A; float* o1 = B; float* o2 = C;
    int dim = F, L = (int)n;
    for (int s = 0; s < 4; ++s) {
        const int r = c.ratios[s], hid = dim / c.compress;
        // MimiResnetBlock: x + conv1(ELU(conv3(ELU(x))))
        hipLaunchKernelGGL(k_mimi_elu, dim3((unsigned)(((size_t)dim * L + 255) / 256)), dim3(256), 0, st, x, o1, (size_t)dim * L);
        M_HIP(m_run_conv(e, e->res1[s], o1, o2, L, /*ELU*/ 6));                         // o2 [hid][L]
        M_HIP(m_run_conv(e, e->res2[s], o2, o1, L, 0, x));                              // o1 = x + conv1(.)  [dim][L]
        (void)hid;
        // ELU + strided conv as fold + 2-tap conv
        const int Lf = m_ceil_div(L, r);
        hipLaunchKernelGGL(k_mimi_fold, dim3((Lf + 255) / 256, dim * r), dim3(256), 0, st, o1, o2, L, r, Lf, 1, 0);       // o2 [dim*r][Lf]
        M_HIP(m_run_conv(e, e->down[s], o2, x, Lf));                                    // x [2dim][Lf]
        dim *= 2; L = Lf;
    }
    hipLaunchKernelGGL(k_mimi_elu, dim3((unsigned)(((size_t)dim * L + 255) / 256)), dim3(256), 0, st, x, o1, (size_t)dim * L);
    float* hs = o2;                                                                     // [H][T25]
    M_HIP(m_run_conv(e, e->last, o1, hs, L));
    Q3I_CHECK(TAP(0, hs, (size_t)H * T25));
    // ---- transformer ----
    float* nrm = tr; float* q = nrm + (size_t)H * tf; float* k = q + (size_t)H * tf; float* v = k + (size_t)H * tf; float* att = v + (size_t)H * tf;
    float* ff = att + (size_t)H * tf;
    const float scale = 1.0f / sqrtf(64.0f);
    for (auto& Lr : e->layers) {
        M_HIP(launch_layernorm_c(hs, Lr.ln1w, Lr.ln1b, nrm, H, T25, c.norm_eps, st));
        M_HIP(m_run_conv(e, Lr.q, nrm, q, T25)); M_HIP(m_run_conv(e, Lr.k, nrm, k, T25)); M_HIP(m_run_conv(e, Lr.v, nrm, v, T25));
        M_HIP(launch_rope_c(q, k, e->cs, e->sn, c.n_heads, 64, T25, st));
        hipLaunchKernelGGL(k_mimi_attn, dim3(T25, c.n_heads), dim3(64), 0, st, q, k, v, att, T25, c.window, scale);
        M_HIP(m_run_conv(e, Lr.o, att, hs, T25, 0, hs, Lr.sa));                          // hs += scale_attn * o_proj(att)
        M_HIP(launch_layernorm_c(hs, Lr.ln2w, Lr.ln2b, nrm, H, T25, c.norm_eps, st));
        M_HIP(m_run_conv(e, Lr.f1, nrm, ff, T25, /*GELU*/ 1));
        M_HIP(m_run_conv(e, Lr.f2, ff, hs, T25, 0, hs, Lr.sm));                          // hs += scale_mlp * fc2(gelu(fc1))
    }
    Q3I_CHECK(TAP(1, hs, (size_t)H * T25));
    // ---- downsample: k = 4, stride 2, replicate padding -> fold with a leading replicate column, drop output column 0 ----
    float* fd = ff + (size_t)c.inter * tf;                                              // [2H][T + 1]
    float* dy = x;                                                                      // [H][T + 1] (x is free now)
    hipLaunchKernelGGL(k_mimi_fold, dim3((T + 1 + 255) / 256, H * 2), dim3(256), 0, st, hs, fd, T25, 2, T, 0, 1);
    M_HIP(m_run_conv(e, e->dsamp, fd, dy, T + 1));
    float* xd = o1;                                                                     // [H][T]
    M_HIP(launch_copy_rows(dy + 1, T + 1, xd, T, H, T, st));
    Q3I_CHECK(TAP(2, xd, (size_t)H * T));
    // ---- split residual vector quantiser ----
    float* proj = fd + (size_t)2 * H * tf;
    for (int g = 0; g < 2; ++g) {
        const int nl = g ? c.n_q - c.n_sem : c.n_sem, q0 = g ? c.n_sem : 0;
        if (nl <= 0) continue;
        M_HIP(m_run_conv(e, e->qproj[g], xd, proj, T));
        hipLaunchKernelGGL(k_mimi_rvq, dim3(T), dim3(256), 0, st, proj, T, e->books[g], nl, c.cb_size, c.cb_dim, e->codes_dev, c.n_q, q0);
    }
    M_HIP(hipGetLastError());
    M_HIP(hipStreamSynchronize(st));
    M_HIP(q3_hipMemcpy(codes_host, e->codes_dev, (size_t)T * c.n_q * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}
