//! The reference's usage example (examples/tts.rs of TrevorS/qwen3-tts-rs), call for call, against this crate:
//! basic synthesis, voice / language selection, custom options, voice cloning (x-vector and ICL), streaming.
//!   Q3TTS_LIB_DIR=../qwen3_tts_rs_amd cargo run --example tts -- --model-dir path/to/model
use anyhow::Result;
use qwen3_tts::{AudioBuffer, Language, Qwen3TTS, Speaker, SynthesisOptions};
use std::env;

fn main() -> Result<()> {
    let model_dir = env::args().skip_while(|a| a != "--model-dir").nth(1).unwrap_or_else(|| "test_data/model".into());
    let device = qwen3_tts::auto_device()?;
    println!("Loading model from: {model_dir}");
    let model = Qwen3TTS::from_pretrained(&model_dir, device)?;

    // 1. basic synthesis (default voice: Ryan, English)
    let audio = model.synthesize("Hello from Rust!", None)?;
    audio.save("output_basic.wav")?;
    println!("Basic: {:.2}s -> output_basic.wav", audio.duration());

    // 2. speaker and language
    let audio = model.synthesize_with_voice("This uses a different voice.", Speaker::Serena, Language::English, None)?;
    audio.save("output_serena.wav")?;
    println!("Serena: {:.2}s -> output_serena.wav", audio.duration());

    // 3. custom generation options
    let options = SynthesisOptions { temperature: 0.9, top_k: 30, max_length: 512, ..Default::default() };
    let audio = model.synthesize_with_voice("Custom sampling parameters.", Speaker::Ryan, Language::English, Some(options))?;
    audio.save("output_custom.wav")?;
    println!("Custom: {:.2}s -> output_custom.wav", audio.duration());

    // 4. voice cloning
    if model.supports_voice_cloning() {
        let ref_audio = AudioBuffer::load("examples/data/clone_2.wav")?;
        let prompt = model.create_voice_clone_prompt(&ref_audio, None)?;                     // x_vector_only
        let audio = model.synthesize_voice_clone("We choose to go to the Moon in this decade.", &prompt, Language::English, None)?;
        audio.save("output_clone.wav")?;
        println!("Clone (x-vector): {:.2}s -> output_clone.wav", audio.duration());
        if model.has_speech_encoder() {                                                      // ICL: embedding + reference codes + transcript
            let ref_text = "Okay. Yeah. I resent you. I love you. I respect you. But you know what? You blew it! And thanks to you.";
            let prompt = model.create_voice_clone_prompt(&ref_audio, Some(ref_text))?;
            let audio = model.synthesize_voice_clone("We choose to go to the Moon in this decade.", &prompt, Language::English, None)?;
            audio.save("output_clone_icl.wav")?;
            println!("Clone (ICL): {:.2}s -> output_clone_icl.wav", audio.duration());
        }
    } else {
        println!("Skipping voice cloning (no speaker encoder in this model)");
    }

    // 5. streaming synthesis
    let options = SynthesisOptions { chunk_frames: 10, ..Default::default() };
    let mut total_samples = 0usize;
    for (i, chunk) in model.synthesize_streaming("Streaming output, chunk by chunk.", Speaker::Ryan, Language::English, options)?.enumerate() {
        let audio = chunk?;
        total_samples += audio.len();
        println!("  chunk {i}: {} samples ({:.2}s)", audio.len(), audio.duration());
    }
    println!("Streaming total: {:.2}s", total_samples as f32 / 24000.0);
    Ok(())
}
