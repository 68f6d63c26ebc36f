"""`python -m qwen3_tts_rs_amd.cli` — the flag surface of the reference's `generate_audio` binary
(src/bin/generate_audio.rs:30-120, README.md:362-380) on the MI355X engine: synthesize one utterance, write the WAV
(PCM16, audio/io.rs:143-165) and the dump files the reference writes (codes_seed{S}_frames{N}.bin i64 LE,
audio_seed{S}_frames{N}.bin f32 LE, metadata_seed{S}_frames{N}.json; generate_audio.rs:724-813).

`--compare --reference-dir DIR` is the reference's own pin hook (generate_audio.rs:69-75, 549-553, 816-931): EOS is
switched off so that both engines run exactly --frames frames, and this run's codes / PCM are compared with the
codes_seed{S}_frames{N}.bin / audio_seed{S}_frames{N}.bin dumps found in DIR — produced by the reference binary (or its
Python upstream) for the same text, seed and sampling flags. tests/PIN_WITH_REFERENCE.md has the exact commands.

Differences, all explicit: `--synthetic {tiny,0.6b,1.7b}` runs without a checkpoint (seeded random weights — there is
no network here); `--token-ids` / `--instruct-ids` bypass the tokenizer; `--ref-audio` runs the speaker encoder (and, with
`--ref-text`, the speech encoder for ICL) of a Base checkpoint on the GPU — `--xvector-npy` / `--ref-codes-bin` inject
precomputed ones instead."""
import argparse
import json
import os
import sys
import time

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="qwen3_tts_rs_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("-t", "--text", default="Hello")
    ap.add_argument("-s", "--seed", type=int, default=42)
    ap.add_argument("-f", "--frames", type=int, default=2048, help="max frames (~164 s); generation stops at EOS")
    ap.add_argument("-d", "--duration", type=float, default=None, help="max duration in seconds (overrides --frames): frames = duration * 12.5")
    ap.add_argument("--temperature", type=float, default=0.7)
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--top-p", type=float, default=0.9)
    ap.add_argument("--repetition-penalty", type=float, default=1.05)
    ap.add_argument("-m", "--model-dir", default="test_data/model")
    ap.add_argument("-o", "--output-dir", default="test_data/rust_audio")
    ap.add_argument("-c", "--compare", action="store_true", help="compare with reference dumps (if they exist); runs exactly --frames frames (no EOS)")
    ap.add_argument("--reference-dir", default="test_data/reference_audio", help="directory holding the reference's codes_/audio_ dumps")
    ap.add_argument("--compare-strict", action="store_true", help="with --compare: exit 1 unless all codes are identical and the PCM is within 1e-3 RMS")
    ap.add_argument("--tokenizer-dir", default=None)
    ap.add_argument("--speaker", default="ryan")
    ap.add_argument("--language", default="english")
    ap.add_argument("--custom-voice", action="store_true", help="accepted for compatibility: the model size always comes from config.json here")
    ap.add_argument("--instruct", default=None, help="voice description (VoiceDesign models)")
    ap.add_argument("--ref-audio", default=None)
    ap.add_argument("--ref-text", default=None)
    ap.add_argument("--x-vector-only", action="store_true")
    ap.add_argument("--output", default=None, help="output WAV path (overrides default naming)")
    ap.add_argument("--device", default="auto", help="auto | hip | hip:N | cuda:N (accepted as an alias)")
    # engine-specific
    ap.add_argument("--synthetic", choices=["tiny", "0.6b", "1.7b"], default=None, help="seeded random weights instead of --model-dir")
    ap.add_argument("--token-ids", default=None, help="comma-separated text token ids (bypasses the tokenizer)")
    ap.add_argument("--instruct-ids", default=None, help="comma-separated instruct token ids")
    ap.add_argument("--xvector-npy", default=None, help="speaker embedding [hidden] f32 (.npy) for voice cloning")
    ap.add_argument("--ref-codes-bin", default=None, help="reference codec frames (codes_*.bin) for ICL voice cloning")
    ap.add_argument("--streaming", action="store_true", help="stream chunks (reports time to first audio)")
    ap.add_argument("--no-eos", action="store_true", help="disable EOS (fixed-length runs on synthetic weights)")
    return ap


def parse_device(s: str) -> int:
    """parse_device (lib.rs:1875-1926): auto / hip / hip:N; cuda[:N] accepted as an alias; cpu / metal are errors here."""
    s = s.lower()
    if s in ("auto", "hip", "cuda", "gpu"):
        return 0
    for p in ("hip:", "cuda:"):
        if s.startswith(p):
            return int(s[len(p):])
    raise ValueError(f"Unsupported device '{s}': this build runs on MI355X only (auto | hip | hip:N)")


def compare_with_reference(reference_dir: str, seed: int, num_frames: int, codes: np.ndarray, audio: np.ndarray, out=print) -> dict:
    """compare_with_reference (generate_audio.rs:816-931): this run's codes / PCM against the dumps
    codes_seed{seed}_frames{num_frames}.bin (i64 LE, frame-major) and audio_seed{seed}_frames{num_frames}.bin (f32 LE) in
    `reference_dir`. Prints the reference's report lines (first five differing indices, max / mean / RMSE, MATCH / CLOSE /
    DIFFERENT at 1e-5 / 1e-3 of max difference) plus what a parity pin needs: the first differing (frame, group) and the
    north-star verdict (all codec ids identical, PCM within 1e-3 RMS). Returns the numbers."""
    from qwen3_tts_rs_amd import api
    rep = {"codes_found": False, "audio_found": False, "codes_match": None, "first_diff": None, "n_diff": None,
           "max_diff": None, "mean_diff": None, "rmse": None, "status": None, "pinned": None}
    mine = np.ascontiguousarray(codes, dtype=np.uint32).reshape(-1, 16)
    cp = os.path.join(reference_dir, f"codes_seed{seed}_frames{num_frames}.bin")
    if os.path.exists(cp):
        ref = api.load_codes_binary(cp)
        rep["codes_found"] = True
        a, b = ref.reshape(-1).astype(np.int64), mine.reshape(-1).astype(np.int64)
        match = a.size == b.size and bool((a == b).all())
        rep["codes_match"] = match
        if match:
            out(f"Codes: MATCH (all {a.size} values identical)")
            rep["n_diff"] = 0
        else:
            out("Codes: MISMATCH"); out(f"  Reference: {a.size} values"); out(f"  This run:  {b.size} values")
            n = min(a.size, b.size)
            bad = np.nonzero(a[:n] != b[:n])[0]
            for i in bad[:5]:
                out(f"  Index {int(i)} (frame {int(i) // 16}, group {int(i) % 16}): Reference={int(a[i])}, This run={int(b[i])}")
            if bad.size > 5:
                out(f"  ... and {bad.size - 5} more differences")
            out(f"  Total differences: {bad.size}")
            rep["n_diff"] = int(bad.size)
            if bad.size:
                rep["first_diff"] = (int(bad[0]) // 16, int(bad[0]) % 16)
                out(f"  First divergence: frame {rep['first_diff'][0]}, group {rep['first_diff'][1]} "
                    f"(group 0 = talker sample, 1..15 = code-predictor argmax; every later frame depends on it)")
    else:
        out(f"Codes: reference not found at {cp!r}")
    ap = os.path.join(reference_dir, f"audio_seed{seed}_frames{num_frames}.bin")
    if os.path.exists(ap):
        ref = api.load_audio_binary(ap)
        rep["audio_found"] = True
        mine_a = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
        n = min(ref.size, mine_a.size)
        if n > 0:
            d = np.abs(ref[:n].astype(np.float64) - mine_a[:n].astype(np.float64))
            rep["max_diff"], rep["mean_diff"], rep["rmse"] = float(d.max()), float(d.mean()), float(np.sqrt(np.mean(d * d)))
            rep["status"] = "MATCH" if rep["max_diff"] < 1e-5 else "CLOSE" if rep["max_diff"] < 1e-3 else "DIFFERENT"
            out(f"\nAudio comparison ({n} samples):"); out(f"  Reference samples: {ref.size}"); out(f"  This run samples:  {mine_a.size}")
            out(f"  Max difference: {rep['max_diff']:.6f}"); out(f"  Mean difference: {rep['mean_diff']:.6f}"); out(f"  RMSE: {rep['rmse']:.6f}")
            out(f"  Status: {rep['status']} (max diff " + ("< 1e-5)" if rep["status"] == "MATCH" else "< 1e-3)" if rep["status"] == "CLOSE" else ">= 1e-3)"))
    else:
        out(f"Audio: reference not found at {ap!r}")
    mp = os.path.join(reference_dir, f"metadata_seed{seed}_frames{num_frames}.json")
    if os.path.exists(mp):
        with open(mp) as f:
            out(f"\nReference metadata: {json.load(f)}")
    if rep["codes_found"] and rep["audio_found"] and rep["rmse"] is not None:
        same_len = rep["codes_match"] is True
        rep["pinned"] = bool(same_len and rep["rmse"] <= 1e-3)
        out(f"\nParity (north star: codec ids bit-exact, PCM within 1e-3 RMS): {'PINNED' if rep['pinned'] else 'NOT pinned'}")
    return rep


def max_frames_from_args(a) -> int:
    return int(a.duration * 12.5) if a.duration is not None else a.frames


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    import qwen3_tts_rs_amd as q
    from qwen3_tts_rs_amd import api
    from qwen3_tts_rs_amd.text import TextTokenizer
    # flag validation of generate_audio.rs:163-210
    if a.instruct and a.ref_audio:
        print("error: --instruct and --ref-audio are mutually exclusive.\n  --instruct is for VoiceDesign models (text-described voices).\n"
              "  --ref-audio is for Base models (voice cloning from reference audio).", file=sys.stderr)
        return 2
    if a.ref_text and not (a.ref_audio or a.xvector_npy):
        print("error: --ref-text requires --ref-audio (reference text is the transcript of the reference audio for ICL voice cloning)", file=sys.stderr)
        return 2
    if a.x_vector_only and not (a.ref_audio or a.xvector_npy):
        print("error: --x-vector-only requires --ref-audio (x_vector_only is a voice cloning mode)", file=sys.stderr)
        return 2
    if a.x_vector_only and a.ref_text:
        print("error: --x-vector-only and --ref-text are contradictory.\n  x_vector_only uses only the speaker embedding (no ICL).", file=sys.stderr)
        return 2
    dev = parse_device(a.device)
    speaker, language = q.Speaker.from_str(a.speaker), q.Language.from_str(a.language)
    frames = max_frames_from_args(a)
    t0 = time.time()
    if a.synthetic:
        cfg = {"tiny": q.tiny, "0.6b": q.qwen3_tts_0_6b, "1.7b": q.qwen3_tts_1_7b}[a.synthetic]()
        model = q.Qwen3TTS.from_synthetic(cfg, device=dev)
        if a.ref_audio:      # a Base-style synthetic model: seeded ECAPA-TDNN of the matching embedding width
            scfg = q.tiny_speaker_config(cfg.hidden) if a.synthetic == "tiny" else q.SpeakerEncoderConfig(enc_dim=cfg.hidden)
            model.attach_speaker_encoder(q.SpeakerEncoder.from_synthetic(scfg, device=dev))
            if a.ref_text and not a.ref_codes_bin:      # ICL from raw audio: seeded speech encoder (Mimi shapes)
                model.attach_speech_encoder(q.SpeechEncoder.from_synthetic(None, device=dev))     # 16 codebooks, whatever the talker's size
        tok = TextTokenizer.from_pretrained(None, a.tokenizer_dir, allow_stand_in=True)     # --synthetic: the labelled stand-in
    else:
        # a real checkpoint needs its real tokenizer (the reference fails to load without one, text.rs:62-110) unless every
        # piece of text arrives as ids
        need_text = not a.token_ids or (a.instruct and not a.instruct_ids) or bool(a.ref_text)
        try:
            tok = TextTokenizer.from_pretrained(a.model_dir, a.tokenizer_dir)
        except FileNotFoundError as e:
            if need_text:
                print(f"error: {e}", file=sys.stderr)
                return 2
            tok = None
        model = q.Qwen3TTS.from_pretrained(a.model_dir, device=dev)
    print(f"Loaded model in {time.time() - t0:.2f}s ({model.config.name}, type {model.model_type.name if model.model_type else 'unknown'}, "
          f"tokenizer: {tok.kind if tok else 'none (--token-ids)'})")
    ids = [int(x) for x in a.token_ids.split(",")] if a.token_ids else tok.encode(a.text)
    opts = q.SynthesisOptions(max_length=frames, temperature=a.temperature, top_k=a.top_k, top_p=a.top_p,
                              repetition_penalty=a.repetition_penalty, seed=a.seed)
    if a.no_eos or a.compare:          # --compare: don't stop early when comparing with the reference (generate_audio.rs:549-553)
        opts.eos_token_id = None
    utt = q.Utterance(ids, speaker, language, seed=a.seed)
    if a.instruct or a.instruct_ids:
        utt.instruct_ids = [int(x) for x in a.instruct_ids.split(",")] if a.instruct_ids else tok.encode(a.instruct)
    if a.ref_audio:          # run_voice_clone (generate_audio.rs:213-300): speaker embedding from the reference WAV
        ref = api.AudioBuffer.load(a.ref_audio)
        print(f"Reference audio: {a.ref_audio} ({ref.duration():.2f}s, {ref.sample_rate} Hz)")
        print("Mode: ICL (reference codes + text)" if a.ref_text else "Mode: x_vector_only (no reference text)")
        try:
            if a.ref_text and not a.ref_codes_bin and not a.x_vector_only:
                # ICL from raw audio: speaker embedding + codec frames of the reference clip (lib.rs:1132-1190)
                prompt = model.create_voice_clone_prompt(ref, ref_text_ids=tok.encode(a.ref_text))
                utt.xvector, utt.ref_codes, utt.ref_text_ids = prompt.speaker_embedding, prompt.ref_codes, prompt.ref_text_ids
            else:
                utt.xvector = model.create_voice_clone_prompt(ref).speaker_embedding
        except api._lib.Q3Error as e:
            print(f"error: {e}", file=sys.stderr)
            return 2
    elif a.xvector_npy:
        utt.xvector = np.load(a.xvector_npy).astype(np.float32).reshape(-1)
    if utt.xvector is not None:
        if a.ref_codes_bin and not a.x_vector_only:
            utt.ref_codes = api.load_codes_binary(a.ref_codes_bin)
            utt.ref_text_ids = tok.encode(a.ref_text or "")
    print(f"Generating up to {frames} frames...")
    t1 = time.time(); ttfa = None
    if a.streaming:
        ss = api.StreamingSession(model, utt, opts)
        chunks = []
        for c in ss:
            if ttfa is None:
                ttfa = (time.time() - t1) * 1000.0
            chunks.append(c.samples)
        samples = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
        codes = ss._s.codes(0); ss._s.close()
        timing = None
    else:
        s = model.session([utt], opts)
        audio, timing = s.run()
        samples, codes = audio[0].samples, s.codes(0); s.close()
    wall = time.time() - t1
    n = int(codes.shape[0])
    audio = api.AudioBuffer(samples, 24000)
    print(f"Generated: {audio.duration():.2f}s, {len(audio)} samples, {n} frames in {wall * 1e3:.0f} ms (RTF {wall / max(audio.duration(), 1e-9):.3f})"
          + (f", TTFA {ttfa:.1f} ms" if ttfa is not None else ""))
    os.makedirs(a.output_dir, exist_ok=True)
    wav = a.output or os.path.join(a.output_dir, f"audio_seed{a.seed}_frames{n}.wav")
    audio.save(wav)
    api.save_codes_binary(os.path.join(a.output_dir, f"codes_seed{a.seed}_frames{n}.bin"), codes)
    api.save_audio_binary(os.path.join(a.output_dir, f"audio_seed{a.seed}_frames{n}.bin"), samples)
    meta = {"text": a.text, "seed": a.seed, "num_frames": n, "temperature": a.temperature, "top_k": a.top_k, "top_p": a.top_p,
            "input_ids": ids, "codes_shape": [n, 16], "audio_samples": int(len(audio)), "sample_rate": 24000}
    with open(os.path.join(a.output_dir, f"metadata_seed{a.seed}_frames{n}.json"), "w") as f:
        json.dump(meta, f, indent=2)
    print(f"Saved WAV to: {wav}")
    if timing is not None:
        print(f"Stages: prefill {timing.prefill_ms:.1f} ms, generation {timing.generation_ms:.1f} ms ({timing.generation_frames} frames), decode {timing.decode_ms:.1f} ms")
    model.close()
    if a.compare:
        print("\n=== Comparing with reference ===")
        rep = compare_with_reference(a.reference_dir, a.seed, n, codes, samples)
        if a.compare_strict and not rep["pinned"]:
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
