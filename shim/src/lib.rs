//! `qwen3_tts` — the reference crate's public surface (src/lib.rs:107-117 re-exports) implemented over the C ABI of
//! libq3tts.so (include/q3tts.h). Every method cites the reference method it stands in for. Text is tokenized here with the
//! `tokenizers` crate exactly as the reference does (src/tokenizer/text.rs); everything after the token ids runs on the GPU.
use anyhow::{anyhow, bail, Result};
use std::ffi::{c_char, c_void, CStr, CString};
use std::path::Path;

pub mod ffi {
    use super::*;
    #[repr(C)] #[derive(Clone, Copy)]
    pub struct Q3Options { pub temperature: f64, pub top_p: f64, pub repetition_penalty: f64, pub seed: u64, pub max_length: i32,
        pub top_k: i32, pub eos_token_id: i32, pub chunk_frames: i32, pub min_new_tokens: i32, pub has_seed: i32 }
    #[repr(C)]
    pub struct Q3Request { pub mode: i32, pub text_ids: *const u32, pub n_text: i32, pub instruct_ids: *const u32, pub n_instruct: i32,
        pub speaker_id: u32, pub language_id: u32, pub xvector: *const f32, pub opts: Q3Options,
        pub ref_codes: *const u32, pub n_ref: i32, pub ref_text_ids: *const u32, pub n_ref_text: i32 }
    /// q3_config (include/q3tts.h): 16 + 2 + 11 + 6 + 2 = 37 four-byte fields, same order
    #[repr(C)] #[derive(Clone, Copy)]
    pub struct Q3Config { pub text_vocab: i32, pub text_dim: i32, pub hidden: i32, pub inter: i32, pub n_layers: i32, pub n_heads: i32, pub n_kv_heads: i32,
        pub head_dim: i32, pub codec_vocab: i32, pub cp_hidden: i32, pub cp_inter: i32, pub cp_layers: i32, pub cp_heads: i32, pub cp_kv_heads: i32,
        pub cp_vocab: i32, pub n_groups: i32, pub rms_eps: f32, pub rope_theta: f32, pub dec_cb_dim: i32, pub dec_q_dim: i32, pub dec_latent: i32,
        pub dec_hidden: i32, pub dec_layers: i32, pub dec_heads: i32, pub dec_head_dim: i32, pub dec_inter: i32, pub dec_cb_size: i32, pub dec_dim: i32,
        pub dec_up_ratios: [i32; 2], pub dec_up_rates: [i32; 4], pub dec_eps: f32, pub dec_theta: f32 }
    #[repr(C)] #[derive(Default, Clone, Copy)]
    pub struct Q3Timing { pub prefill_ms: f64, pub generation_ms: f64, pub decode_ms: f64, pub generation_frames: i32 }
    #[repr(C)] #[derive(Clone, Copy, Default)]
    pub struct Q3SpkConfig { pub mel_dim: i32, pub enc_dim: i32, pub channels: [i32; 5], pub kernel_sizes: [i32; 5], pub dilations: [i32; 5],
        pub attention_channels: i32, pub res2net_scale: i32, pub se_channels: i32, pub sample_rate: i32 }
    #[repr(C)] #[derive(Clone, Copy, Default)]
    pub struct Q3MimiConfig { pub n_filters: i32, pub hidden: i32, pub ratios: [i32; 4], pub kernel: i32, pub res_kernel: i32, pub last_kernel: i32,
        pub compress: i32, pub n_layers: i32, pub n_heads: i32, pub head_dim: i32, pub inter: i32, pub window: i32, pub cb_size: i32,
        pub cb_dim: i32, pub n_q: i32, pub n_sem: i32, pub norm_eps: f32, pub rope_theta: f32 }
    extern "C" {
        pub fn q3_last_error() -> *const c_char;
        pub fn q3_device_count() -> i32;
        pub fn q3_model_load(model_dir: *const c_char, device: i32, out: *mut *mut c_void, model_type: *mut i32) -> i32;
        pub fn q3_model_free(m: *mut c_void);
        pub fn q3_model_create(cfg: *const Q3Config, device: i32, out: *mut *mut c_void) -> i32;
        pub fn q3_config_default(variant: i32, out: *mut Q3Config) -> i32;
        pub fn q3_model_n_tensors(m: *const c_void) -> i32;
        pub fn q3_model_tensor_info(m: *const c_void, i: i32, name: *mut *const c_char, n: *mut i64, stored_dtype: *mut i32) -> i32;
        pub fn q3_model_set_tensor(m: *mut c_void, name: *const c_char, dtype: i32, data: *const c_void, n: i64) -> i32;
        pub fn q3_model_finalize(m: *mut c_void) -> i32;
        pub fn q3_model_kv_pool_limit(m: *mut c_void, max_pages: i32) -> i32;
        pub fn q3_model_kv_pool_trim(m: *mut c_void, bytes_freed: *mut usize) -> i32;
        pub fn q3_model_set_codec_planes(m: *mut c_void, planes: i32) -> i32;
        pub fn q3_model_kv_pool_info(m: *mut c_void, page_positions: *mut i32, page_bytes: *mut usize, pages_total: *mut i32, pages_in_use: *mut i32, pages_peak: *mut i32) -> i32;
        pub fn q3_codes_to_tensor(frames: *const u32, n_frames: i32, out: *mut i64);
        pub fn q3_session_create(m: *mut c_void, reqs: *const Q3Request, batch: i32, out: *mut *mut c_void) -> i32;
        pub fn q3_session_prefill(s: *mut c_void) -> i32;
        pub fn q3_session_generate(s: *mut c_void, n_frames: i32, use_graph: i32) -> i32;
        pub fn q3_session_run(s: *mut c_void, use_graph: i32, pcm: *mut *mut f32, cap: *const usize, n: *mut usize, t: *mut Q3Timing) -> i32;
        pub fn q3_session_next_chunk(s: *mut c_void, pcm: *mut f32, cap: usize, n: *mut usize, done: *mut i32) -> i32;
        pub fn q3_session_next_chunk_row(s: *mut c_void, b: i32, pcm: *mut f32, cap: usize, n: *mut usize, done: *mut i32) -> i32;
        pub fn q3_session_create_reserved(m: *mut c_void, reqs: *const Q3Request, batch: i32, frame_budget: i32, prompt_budget: i32, out: *mut *mut c_void) -> i32;
        pub fn q3_session_replace(s: *mut c_void, b: i32, req: *const Q3Request) -> i32;
        pub fn q3_batcher_create(m: *mut c_void, slots: i32, frame_budget: i32, prompt_budget: i32, out: *mut *mut c_void) -> i32;
        pub fn q3_batcher_free(b: *mut c_void);
        pub fn q3_batcher_submit(b: *mut c_void, req: *const Q3Request, want_pcm: i32, ticket: *mut i64) -> i32;
        pub fn q3_batcher_step(b: *mut c_void, n_frames: i32, use_graph: i32, n_running: *mut i32, n_queued: *mut i32, n_finished: *mut i32) -> i32;
        pub fn q3_batcher_poll(b: *mut c_void, ticket: i64, state: *mut i32, n_frames: *mut i32, n_samples: *mut usize) -> i32;
        pub fn q3_batcher_fetch(b: *mut c_void, ticket: i64, codes: *mut u32, cap_frames: i32, pcm: *mut f32, cap_samples: usize) -> i32;
        pub fn q3_session_frames(s: *mut c_void, b: i32, n_frames: *mut i32, done: *mut i32) -> i32;
        pub fn q3_session_codes(s: *mut c_void, b: i32, codes: *mut u32, cap_frames: i32, n_frames: *mut i32) -> i32;
        pub fn q3_session_decode(s: *mut c_void, b: i32, f0: i32, f1: i32, pcm: *mut f32, cap: usize, n: *mut usize) -> i32;
        pub fn q3_session_free(s: *mut c_void);
        pub fn q3_decode_codes(m: *mut c_void, frames: *const u32, n_frames: i32, pcm: *mut f32, taps: *mut *mut f32) -> i32;
        pub fn q3_wav_read(path: *const c_char, out: *mut f32, cap: i64, n: *mut i64, rate: *mut u32) -> i32;
        pub fn q3_wav_write_pcm16(path: *const c_char, samples: *const f32, n: i64, rate: u32) -> i32;
        pub fn q3_resample(input: *const f32, n: i64, sr_in: u32, sr_out: u32, out: *mut f32, cap: i64, n_out: *mut i64) -> i32;
        pub fn q3_safetensors_info(path: *const c_char, name: *const c_char, dtype: *mut i32, shape: *mut i64, cap: i32, rank: *mut i32) -> i32;
        pub fn q3_spk_config_from_json(path: *const c_char, out: *mut Q3SpkConfig, present: *mut i32) -> i32;
        pub fn q3_spk_create(cfg: *const Q3SpkConfig, device: i32, out: *mut *mut c_void) -> i32;
        pub fn q3_spk_load_safetensors(enc: *mut c_void, path: *const c_char) -> i32;
        pub fn q3_spk_encode(enc: *mut c_void, samples: *const f32, n: i64, sample_rate: u32, out: *mut f32) -> i32;
        pub fn q3_spk_free(enc: *mut c_void);
        pub fn q3_mimi_config_default(out: *mut Q3MimiConfig) -> i32;
        pub fn q3_mimi_create(cfg: *const Q3MimiConfig, device: i32, out: *mut *mut c_void) -> i32;
        pub fn q3_mimi_load_safetensors(enc: *mut c_void, path: *const c_char) -> i32;
        pub fn q3_mimi_encode(enc: *mut c_void, samples: *const f32, n: i64, sample_rate: u32, codes: *mut u32, cap_frames: i32,
                              n_frames: *mut i32, taps: *mut *mut f32) -> i32;
        pub fn q3_mimi_free(enc: *mut c_void);
    }
}
use ffi::*;

fn check(st: i32) -> Result<()> {
    if st == 0 { Ok(()) } else { bail!("{}", unsafe { CStr::from_ptr(q3_last_error()) }.to_string_lossy()) }
}
fn cstr(s: &str) -> CString { CString::new(s).expect("path / name without interior NUL") }

pub const CODEC_EOS_TOKEN_ID: u32 = 2150;     // lib.rs:1466
pub const SAMPLES_PER_FRAME: usize = 1920;    // lib.rs:1469

/// `Device`: this backend has exactly one kind — an MI355X ordinal. (lib.rs:1854-1926 `auto_device` / `parse_device`)
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Device(pub i32);
pub fn auto_device() -> Result<Device> {
    if unsafe { q3_device_count() } <= 0 { bail!("no HIP device visible: the MI355X backend has no CPU fallback") }
    Ok(Device(0))
}
pub fn parse_device(s: &str) -> Result<Device> {
    match s {
        "auto" => auto_device(),
        _ => s.strip_prefix("hip:").or_else(|| s.strip_prefix("cuda:")).and_then(|n| n.parse().ok()).map(Device)
            .ok_or_else(|| anyhow!("unknown device '{s}' (expected auto or hip:N)")),
    }
}

/// talker.rs:59-108
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Language { Chinese, English, Japanese, Korean, German, French, Russian, Portuguese, Spanish, Italian }
impl Language {
    pub fn token_id(self) -> u32 {
        match self { Self::Chinese => 2055, Self::English => 2050, Self::Japanese => 2058, Self::Korean => 2064, Self::German => 2053,
                     Self::French => 2061, Self::Russian => 2069, Self::Portuguese => 2071, Self::Spanish => 2054, Self::Italian => 2070 }
    }
}
/// talker.rs:73-88: case-insensitive names and the two-letter codes
impl std::str::FromStr for Language {
    type Err = anyhow::Error;
    fn from_str(s: &str) -> std::result::Result<Self, Self::Err> {
        const NAMES: [(&str, &str, Language); 10] = [("english", "en", Language::English), ("chinese", "zh", Language::Chinese),
            ("japanese", "ja", Language::Japanese), ("korean", "ko", Language::Korean), ("german", "de", Language::German), ("french", "fr", Language::French),
            ("russian", "ru", Language::Russian), ("portuguese", "pt", Language::Portuguese), ("spanish", "es", Language::Spanish), ("italian", "it", Language::Italian)];
        let l = s.to_lowercase();
        NAMES.iter().find(|(n, c, _)| l == *n || l == *c).map(|t| t.2).ok_or_else(|| anyhow!("Unknown language: {}", s))
    }
}
/// talker.rs:111-157
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Speaker { Serena, Vivian, UncleFu, Ryan, Aiden, OnoAnna, Sohee, Eric, Dylan }
impl Speaker {
    pub fn token_id(self) -> u32 {
        match self { Self::Serena => 3066, Self::Vivian => 3065, Self::UncleFu => 3010, Self::Ryan => 3061, Self::Aiden => 2861,
                     Self::OnoAnna => 2873, Self::Sohee => 2864, Self::Eric => 2875, Self::Dylan => 2878 }
    }
}

/// talker.rs:127-143: case-insensitive names, with and without the underscore
impl std::str::FromStr for Speaker {
    type Err = anyhow::Error;
    fn from_str(s: &str) -> std::result::Result<Self, Self::Err> {
        const NAMES: [(&str, Speaker); 11] = [("ryan", Speaker::Ryan), ("serena", Speaker::Serena), ("vivian", Speaker::Vivian), ("aiden", Speaker::Aiden),
            ("uncle_fu", Speaker::UncleFu), ("unclefu", Speaker::UncleFu), ("ono_anna", Speaker::OnoAnna), ("onoanna", Speaker::OnoAnna),
            ("sohee", Speaker::Sohee), ("eric", Speaker::Eric), ("dylan", Speaker::Dylan)];
        let l = s.to_lowercase();
        NAMES.iter().find(|(n, _)| l == *n).map(|t| t.1).ok_or_else(|| anyhow!("Unknown speaker: {}", s))
    }
}
impl Speaker {
    /// talker.rs:160-171: the language a preset voice was recorded in
    pub fn native_language(self) -> Language {
        match self { Self::Ryan | Self::Aiden => Language::English, Self::OnoAnna => Language::Japanese, Self::Sohee => Language::Korean,
                     Self::Serena | Self::Vivian | Self::UncleFu | Self::Eric | Self::Dylan => Language::Chinese }
    }
}

/// A host tensor as `from_weights` takes it: the reference hands candle `Tensor`s (lib.rs:267-275); without candle the
/// caller passes the checkpoint's bytes as they are stored (bf16 or f32, little-endian) with the element count.
pub struct HostTensor<'a> { pub bf16: bool, pub data: &'a [u8] }
impl HostTensor<'_> { fn elems(&self) -> i64 { (self.data.len() / if self.bf16 { 2 } else { 4 }) as i64 } }

/// `[1, 16, T]` i64, `data[q * T + f]` — what `codes_to_tensor` (lib.rs:1417-1431) builds as a candle tensor
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct CodesTensor { pub data: Vec<i64>, pub shape: [usize; 3] }

/// lib.rs:1417-1431: frames of 16 codebook values -> `[1, 16, T]`; a frame with another length is an error
pub fn codes_to_tensor(codes: &[Vec<u32>]) -> Result<CodesTensor> {
    let t = codes.len();
    if let Some(bad) = codes.iter().find(|f| f.len() != 16) { bail!("codes_to_tensor: frame with {} values (expected 16)", bad.len()) }
    let flat: Vec<u32> = codes.iter().flatten().copied().collect();
    let mut data = vec![0i64; 16 * t];
    if t > 0 { unsafe { q3_codes_to_tensor(flat.as_ptr(), t as i32, data.as_mut_ptr()) } }
    Ok(CodesTensor { data, shape: [1, 16, t] })
}

/// lib.rs:1786-1836 (same fields, same defaults)
#[derive(Clone, Debug)]
pub struct SynthesisOptions { pub max_length: usize, pub temperature: f64, pub top_k: usize, pub top_p: f64, pub repetition_penalty: f64,
    pub eos_token_id: Option<u32>, pub chunk_frames: usize, pub min_new_tokens: usize, pub seed: Option<u64> }
impl Default for SynthesisOptions {
    fn default() -> Self { Self { max_length: 2048, temperature: 0.9, top_k: 50, top_p: 0.9, repetition_penalty: 1.05,
        eos_token_id: Some(CODEC_EOS_TOKEN_ID), chunk_frames: 10, min_new_tokens: 2, seed: None } }
}
impl SynthesisOptions {
    fn to_c(&self) -> Q3Options {
        Q3Options { temperature: self.temperature, top_p: self.top_p, repetition_penalty: self.repetition_penalty, seed: self.seed.unwrap_or(0),
            max_length: self.max_length as i32, top_k: self.top_k as i32, eos_token_id: self.eos_token_id.map(|e| e as i32).unwrap_or(-1),
            chunk_frames: self.chunk_frames as i32, min_new_tokens: self.min_new_tokens as i32, has_seed: self.seed.is_some() as i32 }
    }
}
/// lib.rs:138-147
#[derive(Clone, Copy, Debug, Default)]
pub struct SynthesisTiming { pub prefill_ms: f64, pub generation_ms: f64, pub generation_frames: usize, pub decode_ms: f64 }

/// audio/io.rs:28-34, 106-165
#[derive(Clone, Debug)]
pub struct AudioBuffer { pub samples: Vec<f32>, pub sample_rate: u32 }
impl AudioBuffer {
    pub fn new(samples: Vec<f32>, sample_rate: u32) -> Self { Self { samples, sample_rate } }
    pub fn len(&self) -> usize { self.samples.len() }
    pub fn is_empty(&self) -> bool { self.samples.is_empty() }
    pub fn duration(&self) -> f32 { self.samples.len() as f32 / self.sample_rate as f32 }
    pub fn load<P: AsRef<Path>>(path: P) -> Result<Self> {
        let p = cstr(&path.as_ref().to_string_lossy());
        let (mut n, mut rate) = (0i64, 0u32);
        check(unsafe { q3_wav_read(p.as_ptr(), std::ptr::null_mut(), 0, &mut n, &mut rate) })?;
        let mut v = vec![0f32; n as usize];
        check(unsafe { q3_wav_read(p.as_ptr(), v.as_mut_ptr(), n, &mut n, &mut rate) })?;
        Ok(Self { samples: v, sample_rate: rate })
    }
    pub fn save<P: AsRef<Path>>(&self, path: P) -> Result<()> {
        let p = cstr(&path.as_ref().to_string_lossy());
        check(unsafe { q3_wav_write_pcm16(p.as_ptr(), self.samples.as_ptr(), self.samples.len() as i64, self.sample_rate) })
    }
    fn resampled_24k(&self) -> Result<AudioBuffer> {              // audio::resample_to_24k (audio/resample.rs:174-176)
        if self.sample_rate == 24000 { return Ok(self.clone()) }
        let mut n = 0i64;
        check(unsafe { q3_resample(self.samples.as_ptr(), self.samples.len() as i64, self.sample_rate, 24000, std::ptr::null_mut(), 0, &mut n) })?;
        let mut out = vec![0f32; n as usize];
        check(unsafe { q3_resample(self.samples.as_ptr(), self.samples.len() as i64, self.sample_rate, 24000, out.as_mut_ptr(), n, &mut n) })?;
        Ok(AudioBuffer { samples: out, sample_rate: 24000 })
    }
}

pub type FrameCodes = Vec<Vec<u32>>;                            // lib.rs:121
/// lib.rs:123-134
pub struct VoiceClonePrompt { pub speaker_embedding: Vec<f32>, pub ref_codes: Option<Vec<u32>> /* [T][16] */, pub ref_text_ids: Option<Vec<u32>> }

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ModelType { Base, CustomVoice, VoiceDesign }           // config.rs:176-194

/// lib.rs:154-173
pub struct Qwen3TTS {
    model: *mut c_void, spk: *mut c_void, speech: *mut c_void, spk_dim: usize,
    tokenizer: tokenizers::Tokenizer, model_type: Option<ModelType>, device: Device,
}
unsafe impl Send for Qwen3TTS {}
impl Drop for Qwen3TTS {
    fn drop(&mut self) { unsafe { if !self.speech.is_null() { q3_mimi_free(self.speech) } if !self.spk.is_null() { q3_spk_free(self.spk) } q3_model_free(self.model) } }
}

struct Req<'a> { mode: i32, text: &'a [u32], instruct: &'a [u32], speaker: u32, language: u32, xvector: Option<&'a [f32]>,
                 ref_codes: Option<&'a [u32]>, ref_text: Option<&'a [u32]>, opts: &'a SynthesisOptions }
struct Session(*mut c_void);
impl Drop for Session { fn drop(&mut self) { unsafe { q3_session_free(self.0) } } }

impl Qwen3TTS {
    /// lib.rs:183-261: config.json + model.safetensors + speech_tokenizer/model.safetensors (q3_model_load), tokenizer.json,
    /// the speaker encoder of Base checkpoints (lib.rs:233-249) and the speech encoder (lib.rs:251-258)
    pub fn from_pretrained(model_dir: &str, device: Device) -> Result<Self> {
        Self::from_pretrained_with_tokenizer(model_dir, None, device)
    }
    /// Cap of the model's KV page pool in pages of 128 positions (0 = HBM is the limit). The reference sizes one cache per call and
    /// bails on overflow (kv_cache.rs:293-300); here a request that needs a page beyond the cap fails with the same error before
    /// anything runs.
    pub fn set_kv_pool_limit(&self, max_pages: i32) -> Result<()> { check(unsafe { q3_model_kv_pool_limit(self.model, max_pages) }) }
    /// Slabs of the KV page pool none of whose pages is held go back to the device; returns the bytes freed.
    pub fn kv_pool_trim(&self) -> Result<usize> {
        let mut n: usize = 0;
        check(unsafe { q3_model_kv_pool_trim(self.model, &mut n) })?;
        Ok(n)
    }
    /// bf16 planes per f32 operand in the vocoder's convs: 3 = exact f32 products (default), 2 = hi + mid only (PCM within
    /// 1e-4 RMS of the CPU path instead of 2.5e-5, the vocoder 30 % faster). Token ids do not depend on it.
    pub fn set_codec_planes(&self, planes: i32) -> Result<()> { check(unsafe { q3_model_set_codec_planes(self.model, planes) }) }
    /// (page positions, page bytes, pages held by the pool, pages in use, peak pages in use)
    pub fn kv_pool_info(&self) -> Result<(i32, usize, i32, i32, i32)> {
        let (mut pp, mut pb, mut tot, mut used, mut peak) = (0i32, 0usize, 0i32, 0i32, 0i32);
        check(unsafe { q3_model_kv_pool_info(self.model, &mut pp, &mut pb, &mut tot, &mut used, &mut peak) })?;
        Ok((pp, pb, tot, used, peak))
    }
    fn from_pretrained_no_tokenizer(model_dir: &str, device: Device, tokenizer: tokenizers::Tokenizer) -> Result<Self> {
        let (mut h, mut mt) = (std::ptr::null_mut(), -1i32);
        check(unsafe { q3_model_load(cstr(model_dir).as_ptr(), device.0, &mut h, &mut mt) })?;
        Ok(Self { model: h, spk: std::ptr::null_mut(), speech: std::ptr::null_mut(), spk_dim: 0, tokenizer,
                  model_type: match mt { 0 => Some(ModelType::Base), 1 => Some(ModelType::CustomVoice), 2 => Some(ModelType::VoiceDesign), _ => None }, device })
    }
    /// the speaker encoder of Base checkpoints (lib.rs:233-249) and the speech encoder (lib.rs:251-258)
    fn attach_encoders(&mut self, model_dir: &str) -> Result<()> {
        let me = self; let device = me.device;
        let main = cstr(&format!("{model_dir}/model.safetensors"));
        let mut probe = 0i32;
        if unsafe { q3_safetensors_info(main.as_ptr(), cstr("speaker_encoder.fc.weight").as_ptr(), &mut probe, std::ptr::null_mut(), 0, std::ptr::null_mut()) } == 0 {
            let (mut sc, mut present) = (Q3SpkConfig::default(), 0i32);
            check(unsafe { q3_spk_config_from_json(cstr(&format!("{model_dir}/config.json")).as_ptr(), &mut sc, &mut present) })?;
            check(unsafe { q3_spk_create(&sc, device.0, &mut me.spk) })?;
            check(unsafe { q3_spk_load_safetensors(me.spk, main.as_ptr()) })?;
            me.spk_dim = sc.enc_dim as usize;
        }
        let tok = cstr(&format!("{model_dir}/speech_tokenizer/model.safetensors"));
        if unsafe { q3_safetensors_info(tok.as_ptr(), cstr("encoder.downsample.conv.weight").as_ptr(), &mut probe, std::ptr::null_mut(), 0, std::ptr::null_mut()) } == 0 {
            let mut mc = Q3MimiConfig::default();
            check(unsafe { q3_mimi_config_default(&mut mc) })?;
            check(unsafe { q3_mimi_create(&mc, device.0, &mut me.speech) })?;
            if unsafe { q3_mimi_load_safetensors(me.speech, tok.as_ptr()) } != 0 {       // non-fatal (lib.rs:1362-1388): ICL just stays unavailable
                unsafe { q3_mimi_free(me.speech) };
                me.speech = std::ptr::null_mut();
            }
        }
        Ok(())
    }
    /// lib.rs:192-261: as `from_pretrained`, the tokenizer taken from `tokenizer_id` — a directory holding tokenizer.json or
    /// the file itself; `None` = the model directory. (The reference also accepts a Hugging Face model id and downloads
    /// it; this build has no hub client — SURVEY.md §2 marks it out of scope — so an id that is not a path is an error.)
    pub fn from_pretrained_with_tokenizer(model_dir: &str, tokenizer_id: Option<&str>, device: Device) -> Result<Self> {
        let tok_path = match tokenizer_id {
            None => Path::new(model_dir).join("tokenizer.json"),
            Some(id) if Path::new(id).is_dir() => Path::new(id).join("tokenizer.json"),
            Some(id) if Path::new(id).is_file() => Path::new(id).to_path_buf(),
            Some(id) => bail!("tokenizer '{id}' is neither a directory nor a file (hub ids are not supported by this build)"),
        };
        let tokenizer = tokenizers::Tokenizer::from_file(&tok_path).map_err(|e| anyhow!("Failed to load tokenizer: {e}"))?;
        let mut me = Self::from_pretrained_no_tokenizer(model_dir, device, tokenizer)?;
        me.attach_encoders(model_dir)?;
        Ok(me)
    }
    /// lib.rs:267-275: pre-loaded weight maps (talker + code predictor, decoder) instead of a directory. Model size by
    /// weight inspection as lib.rs:371-381 does it (the talker's hidden width), no speaker / speech encoder attached.
    pub fn from_weights(model_weights: &std::collections::HashMap<String, HostTensor>, decoder_weights: &std::collections::HashMap<String, HostTensor>,
                        text_tokenizer: tokenizers::Tokenizer, device: Device) -> Result<Self> {
        let hidden = model_weights.get("talker.model.norm.weight").map(|t| t.elems()).ok_or_else(|| anyhow!("from_weights: talker.model.norm.weight missing"))?;
        let mut cfg = std::mem::MaybeUninit::<Q3Config>::uninit();
        check(unsafe { q3_config_default(if hidden >= 2048 { 1 } else { 0 }, cfg.as_mut_ptr()) })?;
        let mut h = std::ptr::null_mut();
        check(unsafe { q3_model_create(cfg.as_ptr(), device.0, &mut h) })?;
        let me = Self { model: h, spk: std::ptr::null_mut(), speech: std::ptr::null_mut(), spk_dim: 0, tokenizer: text_tokenizer, model_type: None, device };
        for i in 0..unsafe { q3_model_n_tensors(h) } {
            let (mut name, mut n, mut dt) = (std::ptr::null(), 0i64, 0i32);
            check(unsafe { q3_model_tensor_info(h, i, &mut name, &mut n, &mut dt) })?;
            let key = unsafe { CStr::from_ptr(name) }.to_string_lossy().into_owned();
            let t = model_weights.get(&key).or_else(|| decoder_weights.get(key.strip_prefix("decoder.").map(|_| key.as_str()).unwrap_or(&key)))
                .ok_or_else(|| anyhow!("from_weights: tensor {key} missing"))?;
            if t.elems() != n { bail!("from_weights: {key} has {} elements, expected {n}", t.elems()) }
            check(unsafe { q3_model_set_tensor(h, name, if t.bf16 { 1 } else { 0 }, t.data.as_ptr() as *const c_void, n) })?;
        }
        check(unsafe { q3_model_finalize(h) })?;
        Ok(me)
    }
    pub fn model_type(&self) -> Option<ModelType> { self.model_type }
    pub fn device(&self) -> Device { self.device }
    pub fn supports_voice_cloning(&self) -> bool { matches!(self.model_type, None | Some(ModelType::Base)) }          // lib.rs:390-396
    pub fn supports_preset_speakers(&self) -> bool { matches!(self.model_type, None | Some(ModelType::CustomVoice)) } // lib.rs:398-404
    pub fn supports_voice_design(&self) -> bool { self.model_type == Some(ModelType::VoiceDesign) }                   // lib.rs:409-411
    pub fn has_speech_encoder(&self) -> bool { !self.speech.is_null() }                                               // lib.rs:1049-1051

    fn encode(&self, text: &str) -> Result<Vec<u32>> {
        Ok(self.tokenizer.encode(text, false).map_err(|e| anyhow!("tokenizer: {e}"))?.get_ids().to_vec())
    }
    fn session(&self, r: &Req) -> Result<Session> {
        let req = Q3Request { mode: r.mode, text_ids: r.text.as_ptr(), n_text: r.text.len() as i32,
            instruct_ids: if r.instruct.is_empty() { std::ptr::null() } else { r.instruct.as_ptr() }, n_instruct: r.instruct.len() as i32,
            speaker_id: r.speaker, language_id: r.language, xvector: r.xvector.map(|x| x.as_ptr()).unwrap_or(std::ptr::null()), opts: r.opts.to_c(),
            ref_codes: r.ref_codes.map(|c| c.as_ptr()).unwrap_or(std::ptr::null()), n_ref: r.ref_codes.map(|c| c.len() / 16).unwrap_or(0) as i32,
            ref_text_ids: r.ref_text.map(|c| c.as_ptr()).unwrap_or(std::ptr::null()), n_ref_text: r.ref_text.map(|c| c.len()).unwrap_or(0) as i32 };
        let mut s = std::ptr::null_mut();
        check(unsafe { q3_session_create(self.model, &req, 1, &mut s) })?;
        Ok(Session(s))
    }
    fn run(&self, r: &Req) -> Result<(AudioBuffer, SynthesisTiming)> {
        let s = self.session(r)?;
        let mut pcm = vec![0f32; r.opts.max_length * SAMPLES_PER_FRAME];
        let (mut p, cap, mut n, mut t) = (pcm.as_mut_ptr(), pcm.len(), 0usize, Q3Timing::default());
        check(unsafe { q3_session_run(s.0, 1, &mut p, &cap, &mut n, &mut t) })?;
        pcm.truncate(n);
        Ok((AudioBuffer::new(pcm, 24000), SynthesisTiming { prefill_ms: t.prefill_ms, generation_ms: t.generation_ms,
            generation_frames: t.generation_frames as usize, decode_ms: t.decode_ms }))
    }

    /// lib.rs:416-423
    pub fn synthesize(&self, text: &str, options: Option<SynthesisOptions>) -> Result<AudioBuffer> {
        self.synthesize_with_voice(text, Speaker::Ryan, Language::English, options)
    }
    /// lib.rs:718-784
    pub fn synthesize_with_voice(&self, text: &str, speaker: Speaker, language: Language, options: Option<SynthesisOptions>) -> Result<AudioBuffer> {
        Ok(self.synthesize_with_timing(text, speaker, language, options)?.0)
    }
    /// lib.rs:425-501
    pub fn synthesize_with_timing(&self, text: &str, speaker: Speaker, language: Language, options: Option<SynthesisOptions>) -> Result<(AudioBuffer, SynthesisTiming)> {
        let (o, ids) = (options.unwrap_or_default(), self.encode(text)?);
        self.run(&Req { mode: 0, text: &ids, instruct: &[], speaker: speaker.token_id(), language: language.token_id(), xvector: None,
                        ref_codes: None, ref_text: None, opts: &o })
    }
    /// lib.rs:802-870
    pub fn synthesize_voice_design(&self, text: &str, instruct: &str, language: Language, options: Option<SynthesisOptions>) -> Result<AudioBuffer> {
        let (o, ids, ins) = (options.unwrap_or_default(), self.encode(text)?, self.encode(instruct)?);
        Ok(self.run(&Req { mode: 2, text: &ids, instruct: &ins, speaker: 0, language: language.token_id(), xvector: None, ref_codes: None,
                           ref_text: None, opts: &o })?.0)
    }
    /// lib.rs:1132-1190
    pub fn create_voice_clone_prompt(&self, ref_audio: &AudioBuffer, ref_text: Option<&str>) -> Result<VoiceClonePrompt> {
        if self.spk.is_null() { bail!("Speaker encoder not available. Ensure model weights contain `speaker_encoder.*` keys (only Base models include a speaker encoder).") }
        let a = ref_audio.resampled_24k()?;
        let mut emb = vec![0f32; self.spk_dim];
        check(unsafe { q3_spk_encode(self.spk, a.samples.as_ptr(), a.samples.len() as i64, 24000, emb.as_mut_ptr()) })?;
        let (ref_codes, ref_text_ids) = match ref_text {
            None => (None, None),
            Some(t) => {
                if self.speech.is_null() { bail!("ICL voice cloning requires a speech encoder, but it was not loaded. Ensure the speech tokenizer weights contain encoder keys, or use x_vector_only mode by passing ref_text=None.") }
                let mut nf = 0i32;
                check(unsafe { q3_mimi_encode(self.speech, a.samples.as_ptr(), a.samples.len() as i64, 24000, std::ptr::null_mut(), 0, &mut nf, std::ptr::null_mut()) })?;
                let mut codes = vec![0u32; nf as usize * 16];
                check(unsafe { q3_mimi_encode(self.speech, a.samples.as_ptr(), a.samples.len() as i64, 24000, codes.as_mut_ptr(), nf, &mut nf, std::ptr::null_mut()) })?;
                (Some(codes), Some(self.encode(t)?))
            }
        };
        Ok(VoiceClonePrompt { speaker_embedding: emb, ref_codes, ref_text_ids })
    }
    /// lib.rs:1202-1262
    pub fn synthesize_voice_clone(&self, text: &str, prompt: &VoiceClonePrompt, language: Language, options: Option<SynthesisOptions>) -> Result<AudioBuffer> {
        Ok(self.synthesize_voice_clone_debug(text, prompt, language, options)?.0)
    }
    /// lib.rs:897-1046: also returns the generated frames
    pub fn synthesize_voice_clone_debug(&self, text: &str, prompt: &VoiceClonePrompt, language: Language, options: Option<SynthesisOptions>) -> Result<(AudioBuffer, FrameCodes)> {
        let (o, ids) = (options.unwrap_or_default(), self.encode(text)?);
        let s = self.session(&Req { mode: 1, text: &ids, instruct: &[], speaker: 0, language: language.token_id(), xvector: Some(&prompt.speaker_embedding),
                                    ref_codes: prompt.ref_codes.as_deref(), ref_text: prompt.ref_text_ids.as_deref(), opts: &o })?;
        check(unsafe { q3_session_prefill(s.0) })?;
        check(unsafe { q3_session_generate(s.0, o.max_length as i32, 1) })?;
        let (mut nf, mut done) = (0i32, 0i32);
        check(unsafe { q3_session_frames(s.0, 0, &mut nf, &mut done) })?;
        let mut flat = vec![0u32; nf.max(1) as usize * 16];
        check(unsafe { q3_session_codes(s.0, 0, flat.as_mut_ptr(), nf.max(1), &mut nf) })?;
        let extra = prompt.ref_codes.as_ref().map(|c| c.len() / 16).unwrap_or(0);
        let mut pcm = vec![0f32; (nf as usize + extra) * SAMPLES_PER_FRAME];
        let mut n = 0usize;
        check(unsafe { q3_session_decode(s.0, 0, 0, nf, pcm.as_mut_ptr(), pcm.len(), &mut n) })?;
        pcm.truncate(n);
        Ok((AudioBuffer::new(pcm, 24000), flat[..nf as usize * 16].chunks(16).map(|c| c.to_vec()).collect()))
    }
    /// lib.rs:881-890
    /// lib.rs:873-875
    pub fn codes_to_tensor(&self, codes: &[Vec<u32>]) -> Result<CodesTensor> { codes_to_tensor(codes) }
    pub fn decode_codes(&self, codes: &[Vec<u32>]) -> Result<AudioBuffer> {
        let flat: Vec<u32> = codes.iter().flat_map(|f| f.iter().copied()).collect();
        let mut pcm = vec![0f32; codes.len() * SAMPLES_PER_FRAME];
        check(unsafe { q3_decode_codes(self.model, flat.as_ptr(), codes.len() as i32, pcm.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(AudioBuffer::new(pcm, 24000))
    }
    /// lib.rs:1070-1093
    pub fn synthesize_streaming(&self, text: &str, speaker: Speaker, language: Language, options: SynthesisOptions) -> Result<StreamingSession<'_>> {
        let ids = self.encode(text)?;
        let s = self.session(&Req { mode: 0, text: &ids, instruct: &[], speaker: speaker.token_id(), language: language.token_id(), xvector: None,
                                    ref_codes: None, ref_text: None, opts: &options })?;
        Ok(StreamingSession { s, chunk: options.chunk_frames * SAMPLES_PER_FRAME, done: false, _model: self })
    }
    /// lib.rs:1095-1128
    pub fn synthesize_voice_design_streaming(&self, text: &str, instruct: &str, language: Language, options: SynthesisOptions) -> Result<StreamingSession<'_>> {
        let (ids, ins) = (self.encode(text)?, self.encode(instruct)?);
        let s = self.session(&Req { mode: 2, text: &ids, instruct: &ins, speaker: 0, language: language.token_id(), xvector: None, ref_codes: None,
                                    ref_text: None, opts: &options })?;
        Ok(StreamingSession { s, chunk: options.chunk_frames * SAMPLES_PER_FRAME, done: false, _model: self })
    }
}

/// lib.rs:1484-1782: `Iterator<Item = Result<AudioBuffer>>`, each chunk decoded as the reference does (context-free)
pub struct StreamingSession<'a> { s: Session, chunk: usize, done: bool, _model: &'a Qwen3TTS }
impl StreamingSession<'_> {
    pub fn next_chunk(&mut self) -> Result<Option<AudioBuffer>> {            // lib.rs:1650-1759
        if self.done { return Ok(None) }
        let mut buf = vec![0f32; self.chunk];
        let (mut n, mut d) = (0usize, 0i32);
        check(unsafe { q3_session_next_chunk(self.s.0, buf.as_mut_ptr(), buf.len(), &mut n, &mut d) })?;
        if d != 0 { self.done = true }
        if n == 0 { return Ok(None) }
        buf.truncate(n);
        Ok(Some(AudioBuffer::new(buf, 24000)))
    }
    pub fn frames_generated(&self) -> usize {                               // lib.rs:1761-1764
        let (mut n, mut d) = (0i32, 0i32);
        unsafe { q3_session_frames(self.s.0, 0, &mut n, &mut d) };
        n as usize
    }
    pub fn is_done(&self) -> bool { self.done }                             // lib.rs:1766-1769
}
/// lib.rs:1772-1782, error contract included: a failing `next_chunk` is yielded as `Some(Err(e))` and the session STAYS
/// pollable — `done` is only set by the engine's own end-of-stream flag, so the next call retries the chunk (the C ABI
/// leaves the session's frame state untouched when it returns an error: include/q3tts.h, q3_session_next_chunk);
/// `None` comes only after the flush of the last partial chunk.
impl Iterator for StreamingSession<'_> {
    type Item = Result<AudioBuffer>;
    fn next(&mut self) -> Option<Self::Item> {
        match self.next_chunk() {
            Ok(Some(audio)) => Some(Ok(audio)),
            Ok(None) => None,
            Err(e) => Some(Err(e)),
        }
    }
}

/// Continuous batching (no reference counterpart: the reference synthesizes one utterance per call). A queue of requests
/// through the rows of one engine session — `submit_*` queue, `step` runs one scheduling round, `fetch` returns a
/// finished request's audio. Every request gets the samples of its own `synthesize_*` call.
pub struct Batcher<'a> { b: *mut c_void, model: &'a Qwen3TTS }
impl Drop for Batcher<'_> { fn drop(&mut self) { unsafe { q3_batcher_free(self.b) } } }
impl<'a> Batcher<'a> {
    pub fn new(model: &'a Qwen3TTS, slots: usize, frame_budget: usize, prompt_budget: usize) -> Result<Self> {
        let mut b = std::ptr::null_mut();
        check(unsafe { q3_batcher_create(model.model, slots as i32, frame_budget as i32, prompt_budget as i32, &mut b) })?;
        Ok(Self { b, model })
    }
    fn submit(&self, r: &Req) -> Result<i64> {
        let req = Q3Request { mode: r.mode, text_ids: r.text.as_ptr(), n_text: r.text.len() as i32,
            instruct_ids: if r.instruct.is_empty() { std::ptr::null() } else { r.instruct.as_ptr() }, n_instruct: r.instruct.len() as i32,
            speaker_id: r.speaker, language_id: r.language, xvector: r.xvector.map(|x| x.as_ptr()).unwrap_or(std::ptr::null()), opts: r.opts.to_c(),
            ref_codes: r.ref_codes.map(|c| c.as_ptr()).unwrap_or(std::ptr::null()), n_ref: r.ref_codes.map(|c| c.len() / 16).unwrap_or(0) as i32,
            ref_text_ids: r.ref_text.map(|c| c.as_ptr()).unwrap_or(std::ptr::null()), n_ref_text: r.ref_text.map(|c| c.len()).unwrap_or(0) as i32 };
        let mut t = 0i64;
        check(unsafe { q3_batcher_submit(self.b, &req, 1, &mut t) })?;      // the engine copies the request
        Ok(t)
    }
    /// queue what `synthesize_with_voice` would run; returns the ticket
    pub fn submit_with_voice(&self, text: &str, speaker: Speaker, language: Language, options: Option<SynthesisOptions>) -> Result<i64> {
        let (o, ids) = (options.unwrap_or_default(), self.model.encode(text)?);
        self.submit(&Req { mode: 0, text: &ids, instruct: &[], speaker: speaker.token_id(), language: language.token_id(), xvector: None,
                           ref_codes: None, ref_text: None, opts: &o })
    }
    /// queue what `synthesize_voice_design` would run
    pub fn submit_voice_design(&self, text: &str, instruct: &str, language: Language, options: Option<SynthesisOptions>) -> Result<i64> {
        let (o, ids, ins) = (options.unwrap_or_default(), self.model.encode(text)?, self.model.encode(instruct)?);
        self.submit(&Req { mode: 2, text: &ids, instruct: &ins, speaker: 0, language: language.token_id(), xvector: None, ref_codes: None,
                           ref_text: None, opts: &o })
    }
    /// one scheduling round of up to `n_frames` frames: (rows running, requests queued, tickets finished in this call)
    pub fn step(&self, n_frames: usize) -> Result<(usize, usize, usize)> {
        let (mut a, mut q, mut f) = (0i32, 0i32, 0i32);
        check(unsafe { q3_batcher_step(self.b, n_frames as i32, 1, &mut a, &mut q, &mut f) })?;
        Ok((a as usize, q as usize, f as usize))
    }
    /// `Ok(None)` while the ticket is queued or running; a finished ticket is returned once and released
    pub fn fetch(&self, ticket: i64) -> Result<Option<AudioBuffer>> {
        let (mut st, mut n, mut ns) = (0i32, 0i32, 0usize);
        check(unsafe { q3_batcher_poll(self.b, ticket, &mut st, &mut n, &mut ns) })?;
        if st < 2 { return Ok(None) }
        let mut pcm = vec![0f32; ns];
        check(unsafe { q3_batcher_fetch(self.b, ticket, std::ptr::null_mut(), 0, pcm.as_mut_ptr(), ns) })?;
        Ok(Some(AudioBuffer::new(pcm, 24000)))
    }
}
