// q3_io.cpp — on-disk formats either side of the hot path, host side only:
//   * config.json        → q3_config            (ParsedModelConfig::from_file, src/models/config.rs:238-336)
//   * *.safetensors      → q3_model_set_tensor  (Qwen3TTS::from_pretrained / load_weights, src/lib.rs:180-262, 1390-1396)
//   * PCM16 mono WAV     ← f32 samples          (save_wav / load_wav, src/audio/io.rs:106-165)
//   * codes_*.bin / audio_*.bin dumps           (src/bin/generate_audio.rs:788-813)
// Everything goes through the public C ABI of q3_model.hip (q3_model_create / _set_tensor / _finalize): this
// file never touches device memory itself.
#include "../../include/q3tts.h"
#include "q3_internal.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// a small JSON DOM (config.json and the safetensors header are both plain JSON objects)
// ------------------------------------------------------------------------------------------------
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false;
    double num = 0;
    bool is_int = false;
    int64_t inum = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;   // insertion order kept

    const JVal* get(const char* key) const {
        if (kind != OBJ) return nullptr;
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    // serde_json `v[key].as_u64().unwrap_or(d)` — absent, non-integer or negative → default
    int64_t u64_or(const char* key, int64_t d) const {
        const JVal* v = get(key);
        return (v && v->kind == NUM && v->is_int && v->inum >= 0) ? v->inum : d;
    }
    double f64_or(const char* key, double d) const {
        const JVal* v = get(key);
        return (v && v->kind == NUM) ? v->num : d;
    }
};

struct JParser {
    const char* p; const char* end; std::string err;
    explicit JParser(const char* s, size_t n) : p(s), end(s + n) {}
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char* what) { if (err.empty()) err = what; return false; }
    bool lit(const char* s) {
        const size_t n = strlen(s);
        if ((size_t)(end - p) < n || memcmp(p, s, n) != 0) return fail("bad literal");
        p += n; return true;
    }
    static void utf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    bool hex4(unsigned& v) {
        if (end - p < 4) return fail("short \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string& o) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("short escape");
                const char c = *p++;
                switch (c) {
                    case '"': o += '"'; break; case '\\': o += '\\'; break; case '/': o += '/'; break;
                    case 'b': o += '\b'; break; case 'f': o += '\f'; break; case 'n': o += '\n'; break;
                    case 'r': o += '\r'; break; case 't': o += '\t'; break;
                    case 'u': {
                        unsigned cp; if (!hex4(cp)) return false;
                        if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                            p += 2; unsigned lo; if (!hex4(lo)) return false;
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(o, cp); break;
                    }
                    default: return fail("bad escape");
                }
            } else o += *p++;
        }
        if (p >= end) return fail("unterminated string");
        ++p; return true;
    }
    bool value(JVal& v, int depth) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        const char c = *p;
        if (c == '{') {
            v.kind = JVal::OBJ; ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws(); std::string k; if (!string(k)) return false;
                ws(); if (p >= end || *p != ':') return fail("expected ':'"); ++p;
                v.obj.emplace_back(std::move(k), JVal());
                if (!value(v.obj.back().second, depth + 1)) return false;
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JVal::ARR; ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back(), depth + 1)) return false;
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JVal::STR; return string(v.str); }
        if (c == 't') { v.kind = JVal::BOOL; v.b = true; return lit("true"); }
        if (c == 'f') { v.kind = JVal::BOOL; v.b = false; return lit("false"); }
        if (c == 'n') { v.kind = JVal::NUL; return lit("null"); }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s = p; bool integral = true;
            if (*p == '-') ++p;
            while (p < end && *p >= '0' && *p <= '9') ++p;
            if (p < end && *p == '.') { integral = false; ++p; while (p < end && *p >= '0' && *p <= '9') ++p; }
            if (p < end && (*p == 'e' || *p == 'E')) {
                integral = false; ++p;
                if (p < end && (*p == '+' || *p == '-')) ++p;
                while (p < end && *p >= '0' && *p <= '9') ++p;
            }
            const std::string t(s, p);
            v.kind = JVal::NUM; v.num = strtod(t.c_str(), nullptr); v.is_int = integral;
            if (integral) v.inum = strtoll(t.c_str(), nullptr, 10);
            return true;
        }
        return fail("unexpected character");
    }
};

bool parse_json(const char* s, size_t n, JVal& root, std::string& err) {
    JParser jp(s, n);
    if (!jp.value(root, 0)) { err = jp.err; return false; }
    jp.ws();
    if (jp.p != jp.end) { err = "trailing characters"; return false; }
    return true;
}

bool read_file(const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    const bool ok = !ferror(f);
    fclose(f);
    return ok;
}

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

// read-only mapping of a whole file
struct Mapped {
    const uint8_t* p = nullptr; size_t n = 0; int fd = -1;
    ~Mapped() { if (p) munmap((void*)p, n); if (fd >= 0) close(fd); }
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { p = nullptr; return false; }
        p = (const uint8_t*)m;
        return true;
    }
};

inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {   // subnormal → normalise
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

// TalkerConfig::default (0.6B) / ::custom_voice (1.7B), CodePredictorConfig::default, Decoder12HzConfig::default
// (talker.rs:176-290, code_predictor.rs:48-113, decoder_12hz.rs:14-67)
void config_defaults(int variant, q3_config* c) {
    memset(c, 0, sizeof *c);
    c->text_vocab = 151936; c->text_dim = 2048;
    c->hidden = variant ? 2048 : 1024; c->inter = variant ? 6144 : 3072;
    c->n_layers = 28; c->n_heads = 16; c->n_kv_heads = 8; c->head_dim = 128; c->codec_vocab = 3072;
    c->cp_hidden = 1024; c->cp_inter = 3072; c->cp_layers = 5; c->cp_heads = 16; c->cp_kv_heads = 8;
    c->cp_vocab = 2048; c->n_groups = 16; c->rms_eps = 1e-6f; c->rope_theta = 1e6f;
    c->dec_cb_dim = 256; c->dec_q_dim = 512; c->dec_latent = 1024; c->dec_hidden = 512; c->dec_layers = 8;
    c->dec_heads = 16; c->dec_head_dim = 64; c->dec_inter = 1024; c->dec_cb_size = 2048; c->dec_dim = 1536;
    c->dec_up_ratios[0] = 2; c->dec_up_ratios[1] = 2;
    c->dec_up_rates[0] = 8; c->dec_up_rates[1] = 5; c->dec_up_rates[2] = 4; c->dec_up_rates[3] = 3;
    c->dec_eps = 1e-5f; c->dec_theta = 1e4f;
}

// ------------------------------------------------------------------------------------------------
// safetensors: u64 LE header length, JSON header {name: {dtype, shape, data_offsets:[b,e]}, "__metadata__": {...}},
// then the byte buffer the offsets index into.
// ------------------------------------------------------------------------------------------------
struct StEntry { std::string dtype; std::vector<int64_t> shape; uint64_t b = 0, e = 0; };
struct StFile {
    Mapped map; const uint8_t* data = nullptr; size_t data_len = 0;
    std::vector<std::pair<std::string, StEntry>> entries;
    const StEntry* find(const char* name) const {
        for (auto& kv : entries) if (kv.first == name) return &kv.second;
        return nullptr;
    }
};

q3_status st_open(const char* path, StFile& f) {
    if (!f.map.open(path)) return q3i_set_err(Q3_IO, "Failed to open %s", path);
    if (f.map.n < 8) return q3i_set_err(Q3_IO, "%s: not a safetensors file (shorter than its 8-byte header length)", path);
    uint64_t hl = 0;
    for (int i = 0; i < 8; ++i) hl |= (uint64_t)f.map.p[i] << (8 * i);
    if (hl > f.map.n - 8 || hl > (100u << 20)) return q3i_set_err(Q3_IO, "%s: invalid safetensors header length %llu", path, (unsigned long long)hl);
    JVal root; std::string err;
    if (!parse_json((const char*)f.map.p + 8, (size_t)hl, root, err) || root.kind != JVal::OBJ)
        return q3i_set_err(Q3_IO, "%s: invalid safetensors header JSON (%s)", path, err.c_str());
    f.data = f.map.p + 8 + hl; f.data_len = f.map.n - 8 - (size_t)hl;
    for (auto& kv : root.obj) {
        if (kv.first == "__metadata__") continue;
        const JVal& t = kv.second;
        const JVal *dt = t.get("dtype"), *sh = t.get("shape"), *off = t.get("data_offsets");
        if (!dt || dt->kind != JVal::STR || !sh || sh->kind != JVal::ARR || !off || off->kind != JVal::ARR || off->arr.size() != 2)
            return q3i_set_err(Q3_IO, "%s: malformed header entry for tensor %s", path, kv.first.c_str());
        StEntry e; e.dtype = dt->str;
        for (auto& d : sh->arr) {
            if (d.kind != JVal::NUM || !d.is_int || d.inum < 0) return q3i_set_err(Q3_IO, "%s: bad shape for tensor %s", path, kv.first.c_str());
            e.shape.push_back(d.inum);
        }
        if (!off->arr[0].is_int || !off->arr[1].is_int || off->arr[0].inum < 0 || off->arr[1].inum < off->arr[0].inum ||
            (uint64_t)off->arr[1].inum > f.data_len)
            return q3i_set_err(Q3_IO, "%s: data_offsets of tensor %s fall outside the file", path, kv.first.c_str());
        e.b = (uint64_t)off->arr[0].inum; e.e = (uint64_t)off->arr[1].inum;
        f.entries.emplace_back(kv.first, std::move(e));
    }
    return Q3_OK;
}

int dtype_size(const std::string& d) {
    if (d == "F32") return 4;
    if (d == "BF16" || d == "F16") return 2;
    if (d == "F64") return 8;
    return 0;
}

// upload entry `e` of file `f` under `name`; *n_expect = the element count the model wants
q3_status st_upload(q3_model* m, const StFile& f, const char* path, const char* name, const StEntry& e, int64_t n_expect) {
    int64_t n = 1;
    for (int64_t d : e.shape) n *= d;
    const int es = dtype_size(e.dtype);
    if (!es) return q3i_set_err(Q3_UNSUPPORTED, "%s: tensor %s has unsupported dtype %s", path, name, e.dtype.c_str());
    if ((uint64_t)n * (uint64_t)es != e.e - e.b)
        return q3i_set_err(Q3_IO, "%s: tensor %s: shape does not match its byte range", path, name);
    if (n != n_expect)
        return q3i_set_err(Q3_INVALID_ARG, "%s: tensor %s has %lld elements, expected %lld (shape mismatch with config.json?)", path,
                           name, (long long)n, (long long)n_expect);
    const uint8_t* src = f.data + e.b;
    if (e.dtype == "F32") return q3_model_set_tensor(m, name, Q3_DTYPE_F32, src, n);
    if (e.dtype == "BF16") return q3_model_set_tensor(m, name, Q3_DTYPE_BF16, src, n);
    std::vector<float> tmp((size_t)n);
    if (e.dtype == "F16") {
        const uint16_t* h = (const uint16_t*)src;
        for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = f16_to_f32(h[i]);
    } else {   // F64
        for (int64_t i = 0; i < n; ++i) { double d; memcpy(&d, src + 8 * i, 8); tmp[(size_t)i] = (float)d; }
    }
    return q3_model_set_tensor(m, name, Q3_DTYPE_F32, tmp.data(), n);
}

void put_u16(FILE* f, uint16_t v) { uint8_t b[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; fwrite(b, 1, 2, f); }
void put_u32(FILE* f, uint32_t v) { uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; fwrite(b, 1, 4, f); }
uint32_t rd_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd_u16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace

// ------------------------------------------------------------------------------------------------
// config.json
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_config_default(int variant, q3_config* out) {
    if (!out || (variant != 0 && variant != 1)) return q3i_set_err(Q3_INVALID_ARG, "q3_config_default: variant must be 0 (0.6B) or 1 (1.7B)");
    config_defaults(variant, out);
    return Q3_OK;
}

extern "C" q3_status q3_config_from_json(const char* path, q3_config* out, int* model_type) {
    if (!path || !out) return q3i_set_err(Q3_INVALID_ARG, "q3_config_from_json: null argument");
    std::string text;
    if (!read_file(path, text)) return q3i_set_err(Q3_IO, "Failed to read config from %s", path);
    JVal v; std::string err;
    if (!parse_json(text.data(), text.size(), v, err)) return q3i_set_err(Q3_IO, "Failed to parse config from %s (%s)", path, err.c_str());
    static const JVal none;
    const JVal* t = v.get("talker_config"); if (!t) t = &none;
    const JVal* cp = t->get("code_predictor_config"); if (!cp) cp = &none;
    if (model_type) {
        const JVal* mt = v.get("tts_model_type");
        const std::string s = (mt && mt->kind == JVal::STR) ? mt->str : "base";
        *model_type = s == "custom_voice" ? Q3_MODEL_CUSTOM_VOICE : s == "voice_design" ? Q3_MODEL_VOICE_DESIGN : Q3_MODEL_BASE;
    }
    config_defaults(0, out);   // the reference's unwrap_or defaults are the 0.6B shapes
    out->hidden = (int32_t)t->u64_or("hidden_size", 1024);
    out->inter = (int32_t)t->u64_or("intermediate_size", 3072);
    out->n_layers = (int32_t)t->u64_or("num_hidden_layers", 28);
    out->n_heads = (int32_t)t->u64_or("num_attention_heads", 16);
    out->n_kv_heads = (int32_t)t->u64_or("num_key_value_heads", 8);
    out->head_dim = (int32_t)t->u64_or("head_dim", 128);
    out->codec_vocab = (int32_t)t->u64_or("vocab_size", 3072);
    out->text_vocab = (int32_t)t->u64_or("text_vocab_size", 151936);
    out->text_dim = (int32_t)t->u64_or("text_hidden_size", 2048);
    const double t_eps = t->f64_or("rms_norm_eps", 1e-6), t_theta = t->f64_or("rope_theta", 1000000.0);
    out->rms_eps = (float)t_eps; out->rope_theta = (float)t_theta;
    out->cp_hidden = (int32_t)cp->u64_or("hidden_size", 1024);
    out->cp_inter = (int32_t)cp->u64_or("intermediate_size", 3072);
    out->cp_layers = (int32_t)cp->u64_or("num_hidden_layers", 5);
    out->cp_heads = (int32_t)cp->u64_or("num_attention_heads", 16);
    out->cp_kv_heads = (int32_t)cp->u64_or("num_key_value_heads", 8);
    out->cp_vocab = (int32_t)cp->u64_or("vocab_size", 2048);
    out->n_groups = (int32_t)cp->u64_or("num_code_groups", 16);
    // one (eps, theta, head_dim) pair serves talker and code predictor in q3_config; every published variant agrees
    if ((int32_t)cp->u64_or("head_dim", 128) != out->head_dim || cp->f64_or("rms_norm_eps", 1e-6) != t_eps ||
        cp->f64_or("rope_theta", 1000000.0) != t_theta)
        return q3i_set_err(Q3_UNSUPPORTED, "%s: code predictor head_dim/rms_norm_eps/rope_theta differ from the talker's", path);
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// Resampler (audio/resample.rs:17-171). The reference delegates to rubato's asynchronous sinc resampler with
// sinc_len 128, f_cutoff 0.95, BlackmanHarris2 window (resample.rs:83-97); rubato is a Cargo dependency whose sources
// are not available here, so this is a restatement of the published method with those parameters, not of its code:
// a 128-tap Blackman-Harris²-windowed sinc low-pass at 0.95 x the lower Nyquist, evaluated exactly at every output
// instant (f64) instead of interpolating between 128 oversampled tables. Output i sits at input time i / ratio —
// rubato's stream is additionally delayed by sinc_len / 2 input samples and padded to whole 1024-sample chunks
// (resample.rs:118-159), so lengths differ by that padding ("parity unpinned" for this helper).
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_resample(const float* in, int64_t n, uint32_t sr_in, uint32_t sr_out, float* out, int64_t cap, int64_t* n_out) {
    if (!in || n < 0 || !sr_in || !sr_out) return q3i_set_err(Q3_INVALID_ARG, "q3_resample: bad argument");
    const double ratio = (double)sr_out / (double)sr_in;
    const int64_t no = sr_in == sr_out ? n : (int64_t)llround((double)n * ratio);
    if (n_out) *n_out = no;
    if (!out) return Q3_OK;                                   // length query
    if (cap < no) return q3i_set_err(Q3_INVALID_ARG, "q3_resample: output buffer too small (%lld needed)", (long long)no);
    if (sr_in == sr_out) { memcpy(out, in, (size_t)n * sizeof(float)); return Q3_OK; }     // resample.rs:38-40
    const int half = 64;                                      // sinc_len 128
    const double fc = 0.95 * (ratio < 1.0 ? ratio : 1.0);     // cutoff relative to the input Nyquist
    const double a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168, PI = 3.14159265358979323846;
    for (int64_t i = 0; i < no; ++i) {
        const double t = (double)i / ratio;
        const int64_t c = (int64_t)floor(t);
        double acc = 0.0;
        for (int64_t k = c - half + 1; k <= c + half; ++k) {
            if (k < 0 || k >= n) continue;
            const double u = t - (double)k, v = u / (double)half;      // |v| <= 1
            if (v <= -1.0 || v >= 1.0) continue;
            double w = a0 + a1 * cos(PI * v) + a2 * cos(2.0 * PI * v) + a3 * cos(3.0 * PI * v);
            w *= w;
            const double xs = PI * fc * u;
            const double sinc = fabs(xs) < 1e-12 ? 1.0 : sin(xs) / xs;
            acc += (double)in[k] * fc * sinc * w;
        }
        out[i] = (float)acc;
    }
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// speaker encoder: config.json `speaker_encoder_config` (config.rs:100-174, 233) and its tensors
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_spk_config_from_json(const char* path, q3_spk_config* out, int* present) {
    if (!path || !out) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_config_from_json: null argument");
    std::string text;
    if (!read_file(path, text)) return q3i_set_err(Q3_IO, "Failed to read config from %s", path);
    JVal v; std::string err;
    if (!parse_json(text.data(), text.size(), v, err)) return q3i_set_err(Q3_IO, "Failed to parse config from %s (%s)", path, err.c_str());
    Q3I_CHECK(q3_spk_config_default(out));
    const JVal* sc = v.get("speaker_encoder_config");
    if (present) *present = (sc && sc->kind == JVal::OBJ) ? 1 : 0;
    if (!sc || sc->kind != JVal::OBJ) return Q3_OK;
    out->mel_dim = (int32_t)sc->u64_or("mel_dim", out->mel_dim);
    out->enc_dim = (int32_t)sc->u64_or("enc_dim", out->enc_dim);
    out->attention_channels = (int32_t)sc->u64_or("enc_attention_channels", out->attention_channels);
    out->res2net_scale = (int32_t)sc->u64_or("enc_res2net_scale", out->res2net_scale);
    out->se_channels = (int32_t)sc->u64_or("enc_se_channels", out->se_channels);
    out->sample_rate = (int32_t)sc->u64_or("sample_rate", out->sample_rate);
    auto arr5 = [&](const char* key, int32_t* dst) -> q3_status {
        const JVal* a = sc->get(key);
        if (!a) return Q3_OK;                              // serde default
        if (a->kind != JVal::ARR || a->arr.size() != 5) return q3i_set_err(Q3_UNSUPPORTED, "%s: speaker_encoder_config.%s must list 5 blocks", path, key);
        for (int i = 0; i < 5; ++i) {
            if (a->arr[(size_t)i].kind != JVal::NUM || !a->arr[(size_t)i].is_int) return q3i_set_err(Q3_IO, "%s: speaker_encoder_config.%s: not an integer list", path, key);
            dst[i] = (int32_t)a->arr[(size_t)i].inum;
        }
        return Q3_OK;
    };
    Q3I_CHECK(arr5("enc_channels", out->channels));
    Q3I_CHECK(arr5("enc_kernel_sizes", out->kernel_sizes));
    Q3I_CHECK(arr5("enc_dilations", out->dilations));
    return Q3_OK;
}

extern "C" q3_status q3_spk_load_safetensors(q3_speaker_encoder* enc, const char* path) {
    if (!enc || !path) return q3i_set_err(Q3_INVALID_ARG, "q3_spk_load_safetensors: null argument");
    StFile f; Q3I_CHECK(st_open(path, f));
    bool any = false;
    for (auto& kv : f.entries) if (kv.first.rfind("speaker_encoder.", 0) == 0) { any = true; break; }
    if (!any)   // the reference's hint when the encoder is absent (lib.rs:1137-1153)
        return q3i_set_err(Q3_MISSING_WEIGHT, "Speaker encoder not available. Ensure model weights contain `speaker_encoder.*` keys "
                                              "(only Base models include a speaker encoder).");
    const int nt = q3_spk_n_tensors(enc);
    for (int i = 0; i < nt; ++i) {
        const char* name; int64_t n_expect;
        Q3I_CHECK(q3_spk_tensor_info(enc, i, &name, &n_expect));
        const StEntry* e = f.find(name);
        if (!e) return q3i_set_err(Q3_MISSING_WEIGHT, "Missing weight: %s", name);
        int64_t n = 1;
        for (int64_t d : e->shape) n *= d;
        const int es = dtype_size(e->dtype);
        if (!es) return q3i_set_err(Q3_UNSUPPORTED, "%s: tensor %s has unsupported dtype %s", path, name, e->dtype.c_str());
        if ((uint64_t)n * (uint64_t)es != e->e - e->b) return q3i_set_err(Q3_IO, "%s: tensor %s: shape does not match its byte range", path, name);
        if (n != n_expect)
            return q3i_set_err(Q3_INVALID_ARG, "%s: tensor %s has %lld elements, expected %lld (speaker_encoder_config mismatch?)", path, name,
                               (long long)n, (long long)n_expect);
        const uint8_t* src = f.data + e->b;
        if (e->dtype == "F32") { Q3I_CHECK(q3_spk_set_tensor(enc, name, src, Q3_DTYPE_F32, n)); continue; }
        if (e->dtype == "BF16") { Q3I_CHECK(q3_spk_set_tensor(enc, name, src, Q3_DTYPE_BF16, n)); continue; }
        std::vector<float> tmp((size_t)n);
        if (e->dtype == "F16") {
            const uint16_t* h = (const uint16_t*)src;
            for (int64_t j = 0; j < n; ++j) tmp[(size_t)j] = f16_to_f32(h[j]);
        } else {
            for (int64_t j = 0; j < n; ++j) { double d; memcpy(&d, src + 8 * j, 8); tmp[(size_t)j] = (float)d; }
        }
        Q3I_CHECK(q3_spk_set_tensor(enc, name, tmp.data(), Q3_DTYPE_F32, n));
    }
    return q3_spk_finalize(enc);
}

// Encoder12Hz::from_safetensors (encoder_12hz.rs:45-48, 54-71): the `encoder.*` tensors of speech_tokenizer/model.safetensors
extern "C" q3_status q3_mimi_load_safetensors(q3_speech_encoder* enc, const char* path) {
    if (!enc || !path) return q3i_set_err(Q3_INVALID_ARG, "q3_mimi_load_safetensors: null argument");
    StFile f; Q3I_CHECK(st_open(path, f));
    bool any = false;
    for (auto& kv : f.entries) if (kv.first.rfind("encoder.", 0) == 0) { any = true; break; }
    if (!any) return q3i_set_err(Q3_MISSING_WEIGHT, "No encoder keys found (expected keys starting with 'encoder.')");
    const int nt = q3_mimi_n_tensors(enc);
    for (int i = 0; i < nt; ++i) {
        const char* name; int64_t n_expect;
        Q3I_CHECK(q3_mimi_tensor_info(enc, i, &name, &n_expect));
        const StEntry* e = f.find(name);
        if (!e) return q3i_set_err(Q3_MISSING_WEIGHT, "Missing weight: %s", name);
        int64_t n = 1;
        for (int64_t d : e->shape) n *= d;
        const int es = dtype_size(e->dtype);
        if (!es) return q3i_set_err(Q3_UNSUPPORTED, "%s: tensor %s has unsupported dtype %s", path, name, e->dtype.c_str());
        if ((uint64_t)n * (uint64_t)es != e->e - e->b) return q3i_set_err(Q3_IO, "%s: tensor %s: shape does not match its byte range", path, name);
        if (n != n_expect) return q3i_set_err(Q3_INVALID_ARG, "%s: tensor %s has %lld elements, expected %lld", path, name, (long long)n, (long long)n_expect);
        const uint8_t* src = f.data + e->b;
        if (e->dtype == "F32") { Q3I_CHECK(q3_mimi_set_tensor(enc, name, src, Q3_DTYPE_F32, n)); continue; }
        if (e->dtype == "BF16") { Q3I_CHECK(q3_mimi_set_tensor(enc, name, src, Q3_DTYPE_BF16, n)); continue; }
        std::vector<float> tmp((size_t)n);
        if (e->dtype == "F16") {
            const uint16_t* h = (const uint16_t*)src;
            for (int64_t j = 0; j < n; ++j) tmp[(size_t)j] = f16_to_f32(h[j]);
        } else {
            for (int64_t j = 0; j < n; ++j) { double d; memcpy(&d, src + 8 * j, 8); tmp[(size_t)j] = (float)d; }
        }
        Q3I_CHECK(q3_mimi_set_tensor(enc, name, tmp.data(), Q3_DTYPE_F32, n));
    }
    return q3_mimi_finalize(enc);
}

// ------------------------------------------------------------------------------------------------
// safetensors → model
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_model_load_safetensors(q3_model* m, const char* path, int* n_loaded) {
    if (!m || !path) return q3i_set_err(Q3_INVALID_ARG, "q3_model_load_safetensors: null argument");
    StFile f; Q3I_CHECK(st_open(path, f));
    std::map<std::string, const StEntry*> by_name;
    for (auto& kv : f.entries) by_name[kv.first] = &kv.second;
    int loaded = 0;
    const int nt = q3_model_n_tensors(m);
    for (int i = 0; i < nt; ++i) {
        const char* name; int64_t n; int stored;
        Q3I_CHECK(q3_model_tensor_info(m, i, &name, &n, &stored));
        auto it = by_name.find(name);
        if (it == by_name.end()) continue;   // the other file holds it (or q3_model_finalize reports "Missing weight")
        Q3I_CHECK(st_upload(m, f, path, name, *it->second, n));
        ++loaded;
    }
    if (n_loaded) *n_loaded = loaded;
    return Q3_OK;
}

extern "C" q3_status q3_safetensors_info(const char* path, const char* name, int* dtype_out, int64_t* shape, int cap_dims,
                                         int* n_dims) {
    if (!path || !name) return q3i_set_err(Q3_INVALID_ARG, "q3_safetensors_info: null argument");
    StFile f; Q3I_CHECK(st_open(path, f));
    const StEntry* e = f.find(name);
    if (!e) return q3i_set_err(Q3_MISSING_WEIGHT, "Missing weight: %s", name);
    if (dtype_out) *dtype_out = e->dtype == "F32" ? Q3_DTYPE_F32 : e->dtype == "BF16" ? Q3_DTYPE_BF16 : -1;
    if (n_dims) *n_dims = (int)e->shape.size();
    for (int i = 0; i < cap_dims && i < (int)e->shape.size(); ++i) shape[i] = e->shape[(size_t)i];
    return Q3_OK;
}

extern "C" q3_status q3_model_load(const char* model_dir, int device, q3_model** out, int* model_type) {
    if (!model_dir || !out) return q3i_set_err(Q3_INVALID_ARG, "q3_model_load: null argument");
    *out = nullptr;
    const std::string dir(model_dir);
    const std::string model_path = dir + "/model.safetensors";
    if (!file_exists(model_path))
        return q3i_set_err(Q3_IO, "Model weights not found at %s. Please download the model first.", model_path.c_str());
    // speech tokenizer: <dir>/speech_tokenizer/model.safetensors, else the parent directory's (lib.rs:235-253)
    std::string st_path = dir + "/speech_tokenizer/model.safetensors";
    if (!file_exists(st_path)) {
        std::string d = dir;
        while (d.size() > 1 && d.back() == '/') d.pop_back();
        const size_t slash = d.find_last_of('/');
        const std::string parent = slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : d.substr(0, slash));
        st_path = parent + "/speech_tokenizer/model.safetensors";
        if (!file_exists(st_path)) return q3i_set_err(Q3_IO, "Speech tokenizer weights not found");
    }
    q3_config cfg; int mtype = Q3_MODEL_UNKNOWN;
    const std::string cfg_path = dir + "/config.json";
    bool have_cfg = false;
    if (file_exists(cfg_path)) {
        // a config.json that fails to parse falls back to weight inspection (lib.rs:200-215)
        have_cfg = q3_config_from_json(cfg_path.c_str(), &cfg, &mtype) == Q3_OK;
        if (!have_cfg) mtype = Q3_MODEL_UNKNOWN;
    }
    if (!have_cfg) {
        // detect_talker_config (lib.rs:371-381): hidden size of talker.model.norm.weight picks the variant
        int64_t shape[4] = {0, 0, 0, 0}; int nd = 0;
        q3_status st = q3_safetensors_info(model_path.c_str(), "talker.model.norm.weight", nullptr, shape, 4, &nd);
        if (st != Q3_OK) return q3i_set_err(Q3_MISSING_WEIGHT, "Missing talker.model.norm.weight");
        config_defaults(shape[0] == 2048 ? 1 : 0, &cfg);
    }
    q3_model* m = nullptr;
    Q3I_CHECK(q3_model_create(&cfg, device, &m));
    q3_status st = Q3_OK;
    if (device >= 0) {
        st = q3_model_load_safetensors(m, model_path.c_str(), nullptr);
        if (st == Q3_OK) st = q3_model_load_safetensors(m, st_path.c_str(), nullptr);
        if (st == Q3_OK) st = q3_model_finalize(m);
    }
    if (st != Q3_OK) { q3_model_free(m); return st; }
    *out = m;
    if (model_type) *model_type = mtype;
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// WAV (mono PCM16 out; PCM8/16/24/32 + float32 in, channels averaged)
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_pcm16_from_f32(const float* samples, int64_t n, int16_t* out) {
    if ((!samples || !out) && n > 0) return q3i_set_err(Q3_INVALID_ARG, "q3_pcm16_from_f32: null argument");
    for (int64_t i = 0; i < n; ++i) {
        // `(sample.clamp(-1, 1) * 32767.0) as i16` (audio/io.rs:158-160): truncation toward zero, NaN → 0
        float c = samples[i];
        c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);
        const float s = c * 32767.0f;
        out[i] = std::isnan(s) ? (int16_t)0 : (int16_t)s;
    }
    return Q3_OK;
}

extern "C" q3_status q3_wav_write_pcm16(const char* path, const float* samples, int64_t n, uint32_t sample_rate) {
    if (!path || (!samples && n > 0) || n < 0) return q3i_set_err(Q3_INVALID_ARG, "q3_wav_write_pcm16: bad argument");
    if ((uint64_t)n * 2 > 0xFFFFFFFFull - 36) return q3i_set_err(Q3_INVALID_ARG, "q3_wav_write_pcm16: %lld samples do not fit a RIFF file", (long long)n);
    FILE* f = fopen(path, "wb");
    if (!f) return q3i_set_err(Q3_IO, "Failed to create WAV file: %s", path);
    const uint32_t data_bytes = (uint32_t)n * 2;
    fwrite("RIFF", 1, 4, f); put_u32(f, 36 + data_bytes); fwrite("WAVE", 1, 4, f);
    fwrite("fmt ", 1, 4, f); put_u32(f, 16); put_u16(f, 1); put_u16(f, 1); put_u32(f, sample_rate);
    put_u32(f, sample_rate * 2); put_u16(f, 2); put_u16(f, 16);
    fwrite("data", 1, 4, f); put_u32(f, data_bytes);
    std::vector<int16_t> buf(1 << 15);
    for (int64_t i = 0; i < n;) {
        const int64_t c = std::min<int64_t>((int64_t)buf.size(), n - i);
        q3_pcm16_from_f32(samples + i, c, buf.data());
        std::vector<uint8_t> le((size_t)c * 2);
        for (int64_t j = 0; j < c; ++j) { le[(size_t)(2 * j)] = (uint8_t)(uint16_t)buf[(size_t)j]; le[(size_t)(2 * j + 1)] = (uint8_t)((uint16_t)buf[(size_t)j] >> 8); }
        fwrite(le.data(), 1, le.size(), f);
        i += c;
    }
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) return q3i_set_err(Q3_IO, "Failed to write WAV file: %s", path);
    return Q3_OK;
}

extern "C" q3_status q3_wav_read(const char* path, float* out, int64_t cap, int64_t* n_samples, uint32_t* sample_rate) {
    if (!path || !n_samples) return q3i_set_err(Q3_INVALID_ARG, "q3_wav_read: null argument");
    Mapped mp;
    if (!mp.open(path)) return q3i_set_err(Q3_IO, "Failed to open WAV file: %s", path);
    const uint8_t* p = mp.p; const size_t n = mp.n;
    if (n < 12 || memcmp(p, "RIFF", 4) != 0 || memcmp(p + 8, "WAVE", 4) != 0) return q3i_set_err(Q3_IO, "%s: not a RIFF/WAVE file", path);
    size_t off = 12; int fmt = 0, ch = 0, bits = 0; uint32_t rate = 0; const uint8_t* data = nullptr; size_t dlen = 0;
    while (off + 8 <= n) {
        const uint32_t len = rd_u32(p + off + 4);
        const uint8_t* body = p + off + 8;
        const size_t avail = n - off - 8;
        if (memcmp(p + off, "fmt ", 4) == 0) {
            if (len < 16 || avail < 16) return q3i_set_err(Q3_IO, "%s: short fmt chunk", path);
            fmt = rd_u16(body); ch = rd_u16(body + 2); rate = rd_u32(body + 4); bits = rd_u16(body + 14);
            if (fmt == 0xFFFE && len >= 26 && avail >= 26) fmt = rd_u16(body + 24);   // WAVE_FORMAT_EXTENSIBLE sub-format
        } else if (memcmp(p + off, "data", 4) == 0) {
            data = body; dlen = std::min<size_t>(len, avail); break;
        }
        off += 8 + (size_t)len + (len & 1);
    }
    if (!data || !ch) return q3i_set_err(Q3_IO, "%s: missing fmt or data chunk", path);
    const int bps = bits / 8;
    if (!((fmt == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (fmt == 3 && bits == 32)))
        return q3i_set_err(Q3_UNSUPPORTED, "%s: unsupported WAV sample format (tag %d, %d bits)", path, fmt, bits);
    const int64_t frames = (int64_t)(dlen / ((size_t)bps * (size_t)ch));
    *n_samples = frames;
    if (sample_rate) *sample_rate = rate;
    if (!out) return Q3_OK;    // size query
    if (cap < frames) return q3i_set_err(Q3_INVALID_ARG, "q3_wav_read: buffer holds %lld samples, file has %lld", (long long)cap, (long long)frames);
    // (1 << (bits-1)) as f32 (audio/io.rs:122). For 32-bit PCM the reference's i32 shift wraps to -2^31 and flips the
    // sign of every sample; the positive scale is used here.
    const float max_val = ldexpf(1.0f, bits - 1);
    for (int64_t i = 0; i < frames; ++i) {
        float acc = 0.f;
        for (int c = 0; c < ch; ++c) {
            const uint8_t* s = data + ((size_t)i * (size_t)ch + (size_t)c) * (size_t)bps;
            float v;
            if (fmt == 3) memcpy(&v, s, 4);
            else {
                int32_t iv;
                if (bits == 8) iv = (int32_t)s[0] - 128;                 // 8-bit PCM is unsigned on disk
                else if (bits == 16) iv = (int16_t)rd_u16(s);
                else if (bits == 24) iv = ((int32_t)((uint32_t)s[0] << 8 | (uint32_t)s[1] << 16 | (uint32_t)s[2] << 24)) >> 8;
                else iv = (int32_t)rd_u32(s);
                v = (float)iv / max_val;    // v as f32 / (1 << (bits-1)) as f32   (audio/io.rs:121-126)
            }
            acc += v;                       // channels averaged: chunk.iter().sum() / channels (audio/io.rs:131-135)
        }
        out[i] = ch > 1 ? acc / (float)ch : acc;
    }
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// codes_*.bin (i64 LE, frame-major) and audio_*.bin (f32 LE)
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_codes_write_bin(const char* path, const uint32_t* codes, int n_frames, int n_groups) {
    if (!path || (!codes && n_frames > 0) || n_frames < 0 || n_groups <= 0) return q3i_set_err(Q3_INVALID_ARG, "q3_codes_write_bin: bad argument");
    FILE* f = fopen(path, "wb");
    if (!f) return q3i_set_err(Q3_IO, "Failed to create %s", path);
    std::vector<uint8_t> row((size_t)n_groups * 8);
    for (int i = 0; i < n_frames; ++i) {
        for (int g = 0; g < n_groups; ++g) {
            const uint64_t v = codes[(size_t)i * (size_t)n_groups + (size_t)g];   // `code as i64`: zero-extended
            for (int b = 0; b < 8; ++b) row[(size_t)g * 8 + (size_t)b] = (uint8_t)(v >> (8 * b));
        }
        fwrite(row.data(), 1, row.size(), f);
    }
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) return q3i_set_err(Q3_IO, "Failed to write %s", path);
    return Q3_OK;
}

extern "C" q3_status q3_codes_read_bin(const char* path, uint32_t* codes, int cap_frames, int n_groups, int* n_frames) {
    if (!path || !n_frames || n_groups <= 0) return q3i_set_err(Q3_INVALID_ARG, "q3_codes_read_bin: bad argument");
    Mapped mp;
    if (!mp.open(path)) return q3i_set_err(Q3_IO, "Failed to open %s", path);
    if (mp.n % ((size_t)n_groups * 8)) return q3i_set_err(Q3_IO, "%s: size %zu is not a whole number of %d-code frames", path, mp.n, n_groups);
    const int nf = (int)(mp.n / ((size_t)n_groups * 8));
    *n_frames = nf;
    if (!codes) return Q3_OK;
    if (cap_frames < nf) return q3i_set_err(Q3_INVALID_ARG, "q3_codes_read_bin: buffer holds %d frames, file has %d", cap_frames, nf);
    for (size_t i = 0; i < (size_t)nf * (size_t)n_groups; ++i) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v |= (uint64_t)mp.p[i * 8 + (size_t)b] << (8 * b);
        if (v > 0xFFFFFFFFull) return q3i_set_err(Q3_IO, "%s: code %llu at index %zu does not fit u32", path, (unsigned long long)v, i);
        codes[i] = (uint32_t)v;
    }
    return Q3_OK;
}

extern "C" q3_status q3_audio_write_bin(const char* path, const float* samples, int64_t n) {
    if (!path || (!samples && n > 0) || n < 0) return q3i_set_err(Q3_INVALID_ARG, "q3_audio_write_bin: bad argument");
    FILE* f = fopen(path, "wb");
    if (!f) return q3i_set_err(Q3_IO, "Failed to create %s", path);
    if (n) fwrite(samples, 4, (size_t)n, f);   // gfx950 hosts are little-endian x86-64: f32 LE as is
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) return q3i_set_err(Q3_IO, "Failed to write %s", path);
    return Q3_OK;
}

// reader of the same dump (the reference CLI reads its Python counterpart this way: generate_audio.rs:880-886); out == NULL
// or cap == 0: only *n_samples. A file whose size is not a multiple of 4 is rejected.
extern "C" q3_status q3_audio_read_bin(const char* path, float* out, int64_t cap, int64_t* n_samples) {
    if (!path || !n_samples || cap < 0) return q3i_set_err(Q3_INVALID_ARG, "q3_audio_read_bin: bad argument");
    FILE* f = fopen(path, "rb");
    if (!f) return q3i_set_err(Q3_IO, "Failed to open %s", path);
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (bytes < 0 || bytes % 4 != 0) { fclose(f); return q3i_set_err(Q3_IO, "%s: size %ld is not a whole number of f32 samples", path, bytes); }
    const int64_t n = bytes / 4;
    *n_samples = n;
    if (out && cap > 0) {
        if (cap < n) { fclose(f); return q3i_set_err(Q3_INVALID_ARG, "q3_audio_read_bin: buffer holds %lld of %lld samples", (long long)cap, (long long)n); }
        if (n && fread(out, 4, (size_t)n, f) != (size_t)n) { fclose(f); return q3i_set_err(Q3_IO, "Failed to read %s", path); }
    }
    fclose(f);
    return Q3_OK;
}
