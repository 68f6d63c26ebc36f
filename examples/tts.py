#!/usr/bin/env python3
"""Qwen3-TTS usage examples on the MI355X engine — the reference's examples/tts.rs (TrevorS/qwen3-tts-rs) call for call
through the Python host mirror (qwen3_tts_rs_amd.api): basic synthesis, voice / language selection, custom options, voice
cloning (x-vector and ICL) and streaming.

    python examples/tts.py --model-dir path/to/model [--ref-audio examples/data/clone_2.wav] [--out-dir .]

The Rust side of the same call sequence is the reference's own examples/tts.rs built against the shim crate
(shim/Cargo.toml points its [[example]] into a sibling checkout of the reference — no copy is kept here)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import api


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", default="test_data/model")
    ap.add_argument("--ref-audio", default="examples/data/clone_2.wav")
    ap.add_argument("--out-dir", default=".")
    ap.add_argument("--max-length", type=int, default=None, help="cap on generated frames (synthetic checkpoints rarely emit EOS)")
    ap.add_argument("--seed", type=int, default=None, help="sampling seed for every call (default: time-based, as the reference example)")
    a = ap.parse_args(argv)
    out = lambda name: os.path.join(a.out_dir, name)
    cap = {} if a.max_length is None else {"max_length": a.max_length}
    seedkw = {} if a.seed is None else {"seed": a.seed}
    cap.update(seedkw)

    device = q.auto_device()
    print(f"Loading model from: {a.model_dir}")
    model, tok = q.Qwen3TTS.from_pretrained_with_tokenizer(a.model_dir, None, device)

    # 1. basic synthesis (default voice: Ryan, English)
    audio = model.synthesize(tok.encode("Hello from Rust!"), q.SynthesisOptions(**cap) if cap else None)
    audio.save(out("output_basic.wav"))
    print(f"Basic: {audio.duration():.2f}s -> output_basic.wav")

    # 2. choose a speaker and language
    audio = model.synthesize_with_voice(tok.encode("This uses a different voice."), q.Speaker.Serena, q.Language.English,
                                        q.SynthesisOptions(**cap) if cap else None)
    audio.save(out("output_serena.wav"))
    print(f"Serena: {audio.duration():.2f}s -> output_serena.wav")

    # 3. custom generation options
    options = q.SynthesisOptions(temperature=0.9, top_k=30, max_length=a.max_length or 512, **seedkw)
    audio = model.synthesize_with_voice(tok.encode("Custom sampling parameters."), q.Speaker.Ryan, q.Language.English, options)
    audio.save(out("output_custom.wav"))
    print(f"Custom: {audio.duration():.2f}s -> output_custom.wav")

    # 4. voice cloning
    if model.supports_voice_cloning() and model.has_speaker_encoder():
        ref_audio = q.AudioBuffer.load(a.ref_audio)
        prompt = model.create_voice_clone_prompt(ref_audio)                                   # x_vector_only: speaker embedding only
        audio, _ = model.synthesize_voice_clone_prompt(tok.encode("We choose to go to the Moon in this decade."), prompt, q.Language.English,
                                                       q.SynthesisOptions(**cap) if cap else None)
        audio.save(out("output_clone.wav"))
        print(f"Clone (x-vector): {audio.duration():.2f}s -> output_clone.wav")
        if model.has_speech_encoder():                                                        # ICL: embedding + reference codes + transcript
            ref_text = "Okay. Yeah. I resent you. I love you. I respect you. But you know what? You blew it! And thanks to you."
            prompt = model.create_voice_clone_prompt(ref_audio, ref_text_ids=tok.encode(ref_text))
            audio, _ = model.synthesize_voice_clone_prompt(tok.encode("We choose to go to the Moon in this decade."), prompt, q.Language.English,
                                                           q.SynthesisOptions(**cap) if cap else None)
            audio.save(out("output_clone_icl.wav"))
            print(f"Clone (ICL): {audio.duration():.2f}s -> output_clone_icl.wav")
    else:
        print("Skipping voice cloning (no speaker encoder in this model)")

    # 5. streaming synthesis
    options = q.SynthesisOptions(chunk_frames=10, **cap)          # ~800 ms per chunk
    total = 0
    for i, chunk in enumerate(model.synthesize_streaming(tok.encode("Streaming output, chunk by chunk."), q.Speaker.Ryan, q.Language.English, options)):
        total += len(chunk)
        print(f"  chunk {i}: {len(chunk)} samples ({chunk.duration():.2f}s)")
    print(f"Streaming total: {total / 24000.0:.2f}s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
