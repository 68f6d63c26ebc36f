/*
 * q3_oracle_spk.c — CPU F32 restatement of the reference's speaker-embedding path (x-vector voice
 * cloning): 24 kHz audio -> log-mel spectrogram -> ECAPA-TDNN -> [enc_dim] embedding.
 *
 * TEST INFRASTRUCTURE ONLY (see q3_oracle.h). PARITY STATUS: "parity unpinned" numerically — the
 * reference's FFT is rustfft (Cargo dependency, not vendored) and its convolutions are candle's; what is
 * pinned are the reference's own unit-test facts for this path (speaker.rs:402-470 reflect padding /
 * relu / sigmoid / output shape; mel.rs tests: hann window, filterbank shape, frame count), see
 * tests/test_speaker_encoder.py.
 *
 * Follows: src/audio/mel.rs:47-59 (speaker_encoder config), 135-166 (compute_for_speaker_encoder),
 * 168-227 (stft), 229-241 (filterbank apply), 243-318 (slaney mel scale + filterbank), 320-324 (hann);
 * src/models/speaker.rs:24-51 (reflect_pad_1d), 66-106 (ReflectPadConv1d), 113-139 (TimeDelayNetBlock),
 * 149-197 (Res2NetBlock), 205-226 (SqueezeExcitationBlock), 232-272 (SE-Res2Net block),
 * 280-343 (AttentiveStatisticsPooling), 362-469 (SpeakerEncoder::new / encode / forward).
 */
#include "q3_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SPK_MAX_T 64

typedef struct spk_tensor { char name[96]; float* data; int64_t n; } spk_tensor;

struct q3o_spk {
    q3o_spk_config cfg;
    spk_tensor t[SPK_MAX_T * 2];
    int n_t;
};

q3o_spk* q3o_spk_new(const q3o_spk_config* cfg) {
    q3o_spk* s = (q3o_spk*)calloc(1, sizeof(q3o_spk));
    s->cfg = *cfg;
    return s;
}
void q3o_spk_free(q3o_spk* s) {
    if (!s) return;
    for (int i = 0; i < s->n_t; ++i) free(s->t[i].data);
    free(s);
}
int q3o_spk_set_tensor(q3o_spk* s, const char* name, const float* data, int64_t n) {
    if (s->n_t >= SPK_MAX_T * 2) return -1;
    spk_tensor* t = &s->t[s->n_t++];
    snprintf(t->name, sizeof t->name, "%s", name);
    t->n = n; t->data = (float*)malloc((size_t)n * sizeof(float));
    memcpy(t->data, data, (size_t)n * sizeof(float));
    return 0;
}
static const float* spk_get(const q3o_spk* s, const char* name, int64_t n) {
    for (int i = 0; i < s->n_t; ++i)
        if (!strcmp(s->t[i].name, name)) return s->t[i].n == n ? s->t[i].data : NULL;
    return NULL;
}

/* ---- mel front end ---- */

/* mel.rs:320-324 — periodic Hann, f32 arithmetic */
void q3o_hann_window(int len, float* out) {
    const float PI_F = 3.14159265358979323846f;
    for (int i = 0; i < len; ++i) out[i] = 0.5f * (1.0f - cosf(2.0f * PI_F * (float)i / (float)len));
}

/* mel.rs:243-269 — Slaney scale, f32 */
static float hz_to_mel(float f) {
    const float F_SP = 200.0f / 3.0f, MIN_LOG_HZ = 1000.0f, MIN_LOG_MEL = MIN_LOG_HZ / F_SP, LOGSTEP = 0.06875174f;
    return f < MIN_LOG_HZ ? f / F_SP : MIN_LOG_MEL + logf(f / MIN_LOG_HZ) / LOGSTEP;
}
static float mel_to_hz(float m) {
    const float F_SP = 200.0f / 3.0f, MIN_LOG_HZ = 1000.0f, MIN_LOG_MEL = MIN_LOG_HZ / F_SP, LOGSTEP = 0.06875174f;
    return m < MIN_LOG_MEL ? m * F_SP : MIN_LOG_HZ * expf((m - MIN_LOG_MEL) * LOGSTEP);
}
/* mel.rs:271-318 — triangular filters with Slaney area normalisation; out [n_mels][n_fft/2+1] */
void q3o_mel_filterbank(int sample_rate, int n_fft, int n_mels, float fmin, float fmax, float* out) {
    int n_freqs = n_fft / 2 + 1;
    float mel_min = hz_to_mel(fmin), mel_max = hz_to_mel(fmax);
    float* hz = (float*)malloc((size_t)(n_mels + 2) * sizeof(float));
    for (int i = 0; i <= n_mels + 1; ++i) hz[i] = mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(n_mels + 1));
    memset(out, 0, (size_t)n_mels * n_freqs * sizeof(float));
    for (int i = 0; i < n_mels; ++i) {
        float lo = hz[i], ce = hz[i + 1], up = hz[i + 2];
        float* row = out + (size_t)i * n_freqs;
        for (int j = 0; j < n_freqs; ++j) {
            float freq = (float)j * (float)sample_rate / (float)n_fft;
            if (freq >= lo && freq <= ce && ce > lo) row[j] = (freq - lo) / (ce - lo);
            else if (freq > ce && freq <= up && up > ce) row[j] = (up - freq) / (up - ce);
        }
        float bw = hz[i + 2] - hz[i];
        if (bw > 0.0f) { float en = 2.0f / bw; for (int j = 0; j < n_freqs; ++j) row[j] *= en; }
    }
    free(hz);
}

/* mel.rs:168-193 — reflect padding of (n_fft - hop)/2 samples each side, with the reference's index clamps */
int q3o_mel_frames(int n_samples, int n_fft, int hop) {
    int pad = (n_fft - hop) / 2;
    int padded = n_samples + 2 * pad;
    return padded < n_fft ? 0 : (padded - n_fft) / hop + 1;
}

/* compute_for_speaker_encoder (mel.rs:135-166): magnitude sqrt(re^2 + im^2 + 1e-9), mel filterbank,
 * ln(max(., 1e-5)); out [n_mels][T] (the transposed layout the encoder consumes). The DFT is evaluated in
 * double (the reference: rustfft f32 — its rounding is below the tolerance the tests state). */
int q3o_mel_speaker(const float* samples, int n, float* mel, int cap_frames) {
    const int n_fft = 1024, hop = 256, n_mels = 128, sr = 24000, n_freqs = n_fft / 2 + 1;
    if (n < 1) return -1;
    int pad = (n_fft - hop) / 2, np = n + 2 * pad;
    int T = q3o_mel_frames(n, n_fft, hop);
    if (T > cap_frames) return -1;
    float* p = (float*)malloc((size_t)np * sizeof(float));
    int o = 0;
    for (int i = pad; i >= 1; --i) p[o++] = samples[i < n ? i : n - 1];
    memcpy(p + o, samples, (size_t)n * sizeof(float)); o += n;
    for (int i = 0; i < pad; ++i) p[o++] = samples[n >= 2 + i ? n - 2 - i : 0];
    float* win = (float*)malloc(n_fft * sizeof(float));
    q3o_hann_window(n_fft, win);
    float* fb = (float*)malloc((size_t)n_mels * n_freqs * sizeof(float));
    q3o_mel_filterbank(sr, n_fft, n_mels, 0.0f, (float)sr / 2.0f, fb);
    double* cs = (double*)malloc(n_fft * sizeof(double)), *sn = (double*)malloc(n_fft * sizeof(double));
    for (int i = 0; i < n_fft; ++i) { cs[i] = cos(2.0 * M_PI * i / n_fft); sn[i] = sin(2.0 * M_PI * i / n_fft); }
#pragma omp parallel for schedule(static)
    for (int f = 0; f < T; ++f) {
        float buf[1024], mag[513];
        for (int j = 0; j < n_fft; ++j) buf[j] = p[f * hop + j] * win[j];
        for (int k = 0; k < n_freqs; ++k) {
            double re = 0.0, im = 0.0;
            for (int j = 0; j < n_fft; ++j) {
                int idx = (int)(((long)j * k) & (n_fft - 1));
                re += (double)buf[j] * cs[idx]; im -= (double)buf[j] * sn[idx];
            }
            float ref = (float)re, imf = (float)im;
            mag[k] = sqrtf(ref * ref + imf * imf + 1e-9f);
        }
        for (int m = 0; m < n_mels; ++m) {
            const float* row = fb + (size_t)m * n_freqs;
            float acc = 0.0f;
            for (int k = 0; k < n_freqs; ++k) acc += row[k] * mag[k];
            mel[(size_t)m * T + f] = logf(acc > 1e-5f ? acc : 1e-5f);
        }
    }
    free(p); free(win); free(fb); free(cs); free(sn);
    return T;
}

/* ---- ECAPA-TDNN ---- */

/* speaker.rs:24-51 */
void q3o_reflect_pad_1d(const float* x, int C, int T, int pl, int pr, float* out) {
    int Tp = T + pl + pr;
    for (int c = 0; c < C; ++c) {
        const float* xr = x + (size_t)c * T; float* orow = out + (size_t)c * Tp;
        int o = 0;
        for (int i = pl; i >= 1; --i) orow[o++] = xr[i];
        for (int i = 0; i < T; ++i) orow[o++] = xr[i];
        for (int i = 0; i < pr; ++i) orow[o++] = xr[T - 2 - i];
    }
}

/* ReflectPadConv1d::forward (speaker.rs:66-106): "same" length via reflect padding, then a plain conv
 * (stride 1, no padding); act: 0 none, 1 relu */
static void same_conv(const float* x, const float* w, const float* b, float* y, int cin, int cout, int T, int k, int dil, int act) {
    int tot = dil * (k - 1), pl = tot / 2, pr = tot - pl, Tp = T + tot;
    float* xp = (float*)malloc((size_t)cin * Tp * sizeof(float));
    q3o_reflect_pad_1d(x, cin, T, pl, pr, xp);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co) {
        float* yr = y + (size_t)co * T;
        for (int t = 0; t < T; ++t) yr[t] = 0.0f;
        for (int ci = 0; ci < cin; ++ci) {
            const float* xr = xp + (size_t)ci * Tp;
            const float* wr = w + ((size_t)co * cin + ci) * k;
            for (int kk = 0; kk < k; ++kk) {
                float wv = wr[kk]; const float* xs = xr + kk * dil;
                for (int t = 0; t < T; ++t) yr[t] += wv * xs[t];
            }
        }
        float bv = b ? b[co] : 0.0f;
        for (int t = 0; t < T; ++t) { float v = yr[t] + bv; yr[t] = (act == 1 && !(v > 0.0f)) ? 0.0f : v; }
    }
    free(xp);
}

static float sigmoidf_ref(float x) { return 1.0f / (expf(-x) + 1.0f); }   /* speaker.rs:57-61 */

#define GETW(var, nm, cnt) const float* var = spk_get(s, nm, (int64_t)(cnt)); if (!var) { snprintf(q3o_spk_err, sizeof q3o_spk_err, "speaker encoder tensor %s missing or mis-sized", nm); return -1; }
static char q3o_spk_err[160];
const char* q3o_spk_last_error(void) { return q3o_spk_err; }

int q3o_spk_forward(q3o_spk* s, const float* mel, int T, float* out, float** taps) {
    const q3o_spk_config* c = &s->cfg;
    char nm[96], nm2[96];
    int C0 = c->channels[0];
    if (T < 2) { snprintf(q3o_spk_err, sizeof q3o_spk_err, "need at least 2 mel frames"); return -1; }
    float* h = (float*)malloc((size_t)C0 * T * sizeof(float));
    { GETW(w, "speaker_encoder.blocks.0.conv.weight", (size_t)C0 * c->mel_dim * c->kernel_sizes[0]);
      GETW(b, "speaker_encoder.blocks.0.conv.bias", C0);
      same_conv(mel, w, b, h, c->mel_dim, C0, T, c->kernel_sizes[0], c->dilations[0], 1); }
    if (taps && taps[0]) memcpy(taps[0], h, (size_t)C0 * T * sizeof(float));
    int mfa_in = c->channels[1] + c->channels[2] + c->channels[3];
    float* cat = (float*)malloc((size_t)mfa_in * T * sizeof(float));
    int cat_off = 0;
    for (int bi = 1; bi <= 3; ++bi) {
        int C = c->channels[bi], k = c->kernel_sizes[bi], d = c->dilations[bi], sc = c->res2net_scale, ch = C / sc, se = c->se_channels;
        if (C != (bi == 1 ? C0 : c->channels[bi - 1])) { snprintf(q3o_spk_err, sizeof q3o_spk_err, "residual needs equal channel counts"); return -1; }
        float* o1 = (float*)malloc((size_t)C * T * sizeof(float));
        float* o2 = (float*)malloc((size_t)C * T * sizeof(float));
        snprintf(nm, sizeof nm, "speaker_encoder.blocks.%d.tdnn1.conv.weight", bi); snprintf(nm2, sizeof nm2, "speaker_encoder.blocks.%d.tdnn1.conv.bias", bi);
        { GETW(w, nm, (size_t)C * C); GETW(b, nm2, C); same_conv(h, w, b, o1, C, C, T, 1, 1, 1); }
        /* Res2Net (speaker.rs:180-197): chunk 0 passes through; chunk i+1 (+ previous output for i > 0) -> TDNN */
        memcpy(o2, o1, (size_t)ch * T * sizeof(float));
        float* inp = (float*)malloc((size_t)ch * T * sizeof(float));
        for (int i = 0; i < sc - 1; ++i) {
            const float* chunk = o1 + (size_t)(i + 1) * ch * T;
            const float* prev = o2 + (size_t)i * ch * T;
            for (size_t e = 0; e < (size_t)ch * T; ++e) inp[e] = i == 0 ? chunk[e] : chunk[e] + prev[e];
            snprintf(nm, sizeof nm, "speaker_encoder.blocks.%d.res2net_block.blocks.%d.conv.weight", bi, i);
            snprintf(nm2, sizeof nm2, "speaker_encoder.blocks.%d.res2net_block.blocks.%d.conv.bias", bi, i);
            GETW(w, nm, (size_t)ch * ch * k); GETW(b, nm2, ch);
            same_conv(inp, w, b, o2 + (size_t)(i + 1) * ch * T, ch, ch, T, k, d, 1);
        }
        free(inp);
        snprintf(nm, sizeof nm, "speaker_encoder.blocks.%d.tdnn2.conv.weight", bi); snprintf(nm2, sizeof nm2, "speaker_encoder.blocks.%d.tdnn2.conv.bias", bi);
        { GETW(w, nm, (size_t)C * C); GETW(b, nm2, C); same_conv(o2, w, b, o1, C, C, T, 1, 1, 1); }
        /* SE (speaker.rs:218-226) */
        float* sm = (float*)malloc((size_t)C * sizeof(float)), *s1 = (float*)malloc((size_t)se * sizeof(float));
        for (int ci = 0; ci < C; ++ci) { float a = 0.0f; for (int t = 0; t < T; ++t) a += o1[(size_t)ci * T + t]; sm[ci] = a / (float)T; }
        snprintf(nm, sizeof nm, "speaker_encoder.blocks.%d.se_block.conv1.weight", bi); snprintf(nm2, sizeof nm2, "speaker_encoder.blocks.%d.se_block.conv1.bias", bi);
        { GETW(w, nm, (size_t)se * C); GETW(b, nm2, se);
          for (int o = 0; o < se; ++o) { float a = 0.0f; for (int ci = 0; ci < C; ++ci) a += w[(size_t)o * C + ci] * sm[ci]; a += b[o]; s1[o] = a > 0.0f ? a : 0.0f; } }
        snprintf(nm, sizeof nm, "speaker_encoder.blocks.%d.se_block.conv2.weight", bi); snprintf(nm2, sizeof nm2, "speaker_encoder.blocks.%d.se_block.conv2.bias", bi);
        { GETW(w, nm, (size_t)C * se); GETW(b, nm2, C);
          for (int o = 0; o < C; ++o) {
              float a = 0.0f; for (int ci = 0; ci < se; ++ci) a += w[(size_t)o * se + ci] * s1[ci];
              float g = sigmoidf_ref(a + b[o]);
              for (int t = 0; t < T; ++t) { size_t e = (size_t)o * T + t; h[e] = o1[e] * g + h[e]; }     /* out * s + residual */
          } }
        free(sm); free(s1); free(o1); free(o2);
        memcpy(cat + (size_t)cat_off * T, h, (size_t)C * T * sizeof(float)); cat_off += C;
        if (taps && taps[bi]) memcpy(taps[bi], h, (size_t)C * T * sizeof(float));
    }
    free(h);
    int C4 = c->channels[4], A = c->attention_channels;
    float* m = (float*)malloc((size_t)C4 * T * sizeof(float));
    { GETW(w, "speaker_encoder.mfa.conv.weight", (size_t)C4 * mfa_in * c->kernel_sizes[4]); GETW(b, "speaker_encoder.mfa.conv.bias", C4);
      same_conv(cat, w, b, m, mfa_in, C4, T, c->kernel_sizes[4], c->dilations[4], 1); }
    free(cat);
    if (taps && taps[4]) memcpy(taps[4], m, (size_t)C4 * T * sizeof(float));
    /* ASP (speaker.rs:301-343) */
    float* ain = (float*)malloc((size_t)3 * C4 * T * sizeof(float));
    memcpy(ain, m, (size_t)C4 * T * sizeof(float));
    for (int ci = 0; ci < C4; ++ci) {
        const float* r = m + (size_t)ci * T;
        float a = 0.0f; for (int t = 0; t < T; ++t) a += r[t];
        float mean = a / (float)T;
        float q = 0.0f; for (int t = 0; t < T; ++t) { float dd = r[t] - mean; q += dd * dd; }
        float sd = sqrtf(q / (float)T + 1e-5f);
        for (int t = 0; t < T; ++t) { ain[(size_t)(C4 + ci) * T + t] = mean; ain[(size_t)(2 * C4 + ci) * T + t] = sd; }
    }
    float* at = (float*)malloc((size_t)A * T * sizeof(float));
    { GETW(w, "speaker_encoder.asp.tdnn.conv.weight", (size_t)A * 3 * C4); GETW(b, "speaker_encoder.asp.tdnn.conv.bias", A);
      same_conv(ain, w, b, at, 3 * C4, A, T, 1, 1, 1); }
    free(ain);
    for (size_t e = 0; e < (size_t)A * T; ++e) at[e] = tanhf(at[e]);
    float* aw = (float*)malloc((size_t)C4 * T * sizeof(float));
    { GETW(w, "speaker_encoder.asp.conv.weight", (size_t)C4 * A); GETW(b, "speaker_encoder.asp.conv.bias", C4);
      same_conv(at, w, b, aw, A, C4, T, 1, 1, 0); }
    free(at);
    float* pooled = (float*)malloc((size_t)2 * C4 * sizeof(float));
    for (int ci = 0; ci < C4; ++ci) {
        float* a = aw + (size_t)ci * T; const float* r = m + (size_t)ci * T;
        float mx = a[0]; for (int t = 1; t < T; ++t) mx = a[t] > mx ? a[t] : mx;       /* softmax_last_dim: max-subtract, exp, sum, divide */
        float sum = 0.0f; for (int t = 0; t < T; ++t) { a[t] = expf(a[t] - mx); sum += a[t]; }
        for (int t = 0; t < T; ++t) a[t] = a[t] / sum;
        float wm = 0.0f; for (int t = 0; t < T; ++t) wm += r[t] * a[t];
        float wv = 0.0f; for (int t = 0; t < T; ++t) { float dd = r[t] - wm; wv += dd * dd * a[t]; }
        pooled[ci] = wm; pooled[C4 + ci] = sqrtf(wv + 1e-5f);
    }
    free(aw); free(m);
    if (taps && taps[5]) memcpy(taps[5], pooled, (size_t)2 * C4 * sizeof(float));
    { GETW(w, "speaker_encoder.fc.weight", (size_t)c->enc_dim * 2 * C4); GETW(b, "speaker_encoder.fc.bias", c->enc_dim);
      for (int o = 0; o < c->enc_dim; ++o) { float a = 0.0f; for (int ci = 0; ci < 2 * C4; ++ci) a += w[(size_t)o * 2 * C4 + ci] * pooled[ci]; out[o] = a + b[o]; } }
    free(pooled);
    return 0;
}

/* SpeakerEncoder::encode (speaker.rs:431-438) */
int q3o_spk_encode(q3o_spk* s, const float* samples, int n, float* out) {
    int T = q3o_mel_frames(n, 1024, 256);
    if (T < 2) { snprintf(q3o_spk_err, sizeof q3o_spk_err, "reference audio too short"); return -1; }
    float* mel = (float*)malloc((size_t)128 * T * sizeof(float));
    if (q3o_mel_speaker(samples, n, mel, T) != T) { free(mel); return -1; }
    int rc = q3o_spk_forward(s, mel, T, out, NULL);
    free(mel);
    return rc;
}
