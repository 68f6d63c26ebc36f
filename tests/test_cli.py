"""CLI / e2e_bench surface: flag parsing and the tokenizer stand-in on CPU; the end-to-end runs
(synthetic tiny checkpoint) on the GPU."""
import json
import os

import numpy as np
import pytest

from qwen3_tts_rs_amd import api, cli
from qwen3_tts_rs_amd.text import TextTokenizer


def test_cli_flag_surface_and_defaults():
    a = cli.build_parser().parse_args([])
    # README.md:362-380 defaults of the reference CLI
    assert (a.text, a.seed, a.frames, a.temperature, a.top_k, a.top_p, a.repetition_penalty) == ("Hello", 42, 2048, 0.7, 50, 0.9, 1.05)
    assert (a.model_dir, a.speaker, a.language, a.device, a.duration) == ("test_data/model", "ryan", "english", "auto", None)
    a = cli.build_parser().parse_args(["--duration", "4", "--frames", "7"])
    assert cli.max_frames_from_args(a) == 50          # duration * 12.5 overrides --frames (generate_audio.rs:138-144)
    assert cli.parse_device("auto") == 0 and cli.parse_device("hip:3") == 3 and cli.parse_device("cuda:1") == 1
    with pytest.raises(ValueError, match="MI355X"):
        cli.parse_device("cpu")


def test_tokenizer_standin_and_tokenizer_json(tmp_path):
    t = TextTokenizer.from_pretrained(None)
    ids = t.encode("The quick brown fox, transformative!")
    assert t.kind == "synthetic-wordpiece" and ids == t.encode("The quick brown fox, transformative!") and max(ids) < 151643
    assert len(ids) == 4 + 1 + 2 + 1                  # long word splits in two, punctuation is its own piece
    # a real tokenizer.json (WordLevel built with the `tokenizers` package) is picked up from the model directory
    from tokenizers import Tokenizer, models, pre_tokenizers
    tk = Tokenizer(models.WordLevel({"hello": 5, "world": 9, "[UNK]": 0}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    (tmp_path / "m").mkdir()
    tk.save(str(tmp_path / "m" / "tokenizer.json"))
    t2 = TextTokenizer.from_pretrained(str(tmp_path / "m"))
    assert t2.kind == "tokenizer.json" and t2.encode("hello world zzz") == [5, 9, 0]
    # a real model directory WITHOUT a tokenizer is an error (the reference fails to load, text.rs:62-110) — never the stand-in
    (tmp_path / "bare").mkdir()
    with pytest.raises(FileNotFoundError, match="Failed to load tokenizer"):
        TextTokenizer.from_pretrained(str(tmp_path / "bare"))
    assert TextTokenizer.from_pretrained(str(tmp_path / "bare"), allow_stand_in=True).kind == "synthetic-wordpiece"
    assert cli.main(["--model-dir", str(tmp_path / "bare"), "--text", "hello"]) == 2      # exits non-zero before touching the GPU


@pytest.mark.gpu
def test_cli_end_to_end_synthetic(tmp_path):
    out = tmp_path / "o"
    rc = cli.main(["--synthetic", "tiny", "--text", "The quick brown fox", "--frames", "9", "--no-eos", "--seed", "7",
                   "--output-dir", str(out), "--language", "de", "--speaker", "vivian"])
    assert rc == 0
    meta = json.load(open(out / "metadata_seed7_frames9.json"))
    assert meta["num_frames"] == 9 and meta["audio_samples"] == 9 * 1920 and meta["codes_shape"] == [9, 16] and len(meta["input_ids"]) == 4
    codes = np.fromfile(out / "codes_seed7_frames9.bin", dtype="<i8").reshape(9, 16)
    pcm = np.fromfile(out / "audio_seed7_frames9.bin", dtype="<f4")
    assert codes.min() >= 0 and codes[:, 0].max() < 3072 and pcm.shape == (9 * 1920,)
    import wave
    with wave.open(str(out / "audio_seed7_frames9.wav")) as w:
        assert (w.getnframes(), w.getframerate(), w.getsampwidth()) == (9 * 1920, 24000, 2)
    # same seed → same codes (streaming path too)
    rc = cli.main(["--synthetic", "tiny", "--text", "The quick brown fox", "--frames", "9", "--no-eos", "--seed", "7",
                   "--output-dir", str(tmp_path / "o2"), "--language", "de", "--speaker", "vivian", "--streaming"])
    assert rc == 0
    np.testing.assert_array_equal(np.fromfile(tmp_path / "o2" / "codes_seed7_frames9.bin", dtype="<i8").reshape(9, 16), codes)
    # voice cloning from a reference WAV (generate_audio.rs:213-300), x_vector_only; flag rules of :163-210
    t = np.arange(24000 * 2) / 24000.0
    api.save_wav(str(tmp_path / "ref.wav"), (0.4 * np.sin(2 * np.pi * 200 * t) * (0.6 + 0.4 * np.sin(7 * t))).astype(np.float32))
    rc = cli.main(["--synthetic", "tiny", "--text", "The quick brown fox", "--frames", "5", "--no-eos", "--seed", "7",
                   "--output-dir", str(tmp_path / "o3"), "--ref-audio", str(tmp_path / "ref.wav"), "--x-vector-only"])
    assert rc == 0
    c3 = np.fromfile(tmp_path / "o3" / "codes_seed7_frames5.bin", dtype="<i8").reshape(5, 16)
    assert not np.array_equal(c3, codes[:5])          # the speaker embedding conditions the prefill
    # ICL from raw audio (--ref-audio + --ref-text): speaker embedding + codec frames of the clip from the speech encoder
    rc = cli.main(["--synthetic", "tiny", "--text", "The quick brown fox", "--frames", "5", "--no-eos", "--seed", "7",
                   "--output-dir", str(tmp_path / "o4"), "--ref-audio", str(tmp_path / "ref.wav"), "--ref-text", "hello there"])
    assert rc == 0
    c4 = np.fromfile(tmp_path / "o4" / "codes_seed7_frames5.bin", dtype="<i8").reshape(5, 16)
    assert not np.array_equal(c4, c3)                 # the ICL block changes the prefill
    assert cli.main(["--synthetic", "tiny", "--ref-audio", "x.wav", "--instruct", "deep voice"]) == 2
    assert cli.main(["--synthetic", "tiny", "--x-vector-only"]) == 2 and cli.main(["--synthetic", "tiny", "--ref-text", "hi"]) == 2


@pytest.mark.gpu
def test_e2e_bench_report_schema(tmp_path):
    from qwen3_tts_rs_amd import e2e_bench
    p = tmp_path / "r.json"
    assert e2e_bench.main(["--synthetic", "tiny", "--iterations", "2", "--warmup", "0", "--only", "short,medium", "--max-frames", "12",
                           "--json-output", str(p)]) == 0
    rep = json.load(open(p))
    assert set(rep) >= {"device", "model_dir", "iterations", "results"} and [r["label"] for r in rep["results"]] == ["short", "medium"]
    r = rep["results"][0]
    for k in ("label", "text", "word_count", "wall_clock_ms", "wall_clock_stddev_ms", "wall_clock_min_ms", "wall_clock_max_ms",
              "audio_duration_secs", "rtf", "ttfa_ms", "tokens_per_sec", "frames_generated", "peak_memory_mb", "stages"):
        assert k in r
    assert r["word_count"] == 13 and set(r["stages"]) == {"prefill_ms", "generation_ms", "generation_frames", "decode_ms"}
    assert e2e_bench.main(["--synthetic", "tiny", "--iterations", "1", "--warmup", "0", "--only", "short", "--max-frames", "12", "--streaming",
                           "--json-output", str(p)]) == 0
    assert json.load(open(p))["results"][0]["ttfa_ms"] > 0


def test_api_surface_mirrors_the_reference():
    """Every public method of `Qwen3TTS` the reference exports (lib.rs:183-1325, SURVEY.md §8b "Signatures to mirror") has
    a counterpart on the Python host-side mirror; constants and enums carry the reference's values."""
    import qwen3_tts_rs_amd as q
    want = ["from_pretrained", "from_pretrained_with_tokenizer", "from_tensors", "synthesize", "synthesize_with_voice",
            "synthesize_with_timing", "synthesize_voice_design", "create_voice_clone_prompt", "synthesize_voice_clone",
            "synthesize_voice_clone_debug", "synthesize_streaming", "synthesize_voice_design_streaming", "decode_codes",
            "supports_voice_cloning", "supports_preset_speakers", "supports_voice_design", "has_speech_encoder", "device"]
    for name in want:
        assert callable(getattr(q.Qwen3TTS, name)), name
    for name in ("next_chunk", "frames_generated", "is_done", "__iter__"):
        assert hasattr(q.StreamingSession, name), name
    assert q.CODEC_EOS_TOKEN_ID == 2150 and q.SAMPLES_PER_FRAME == 1920                      # lib.rs:1466-1469
    o = q.SynthesisOptions()                                                                  # lib.rs:1786-1836
    assert (o.max_length, o.temperature, o.top_k, o.top_p, o.repetition_penalty, o.eos_token_id, o.chunk_frames, o.min_new_tokens) == \
        (2048, 0.9, 50, 0.9, 1.05, 2150, 10, 2)
    assert q.Speaker.from_str("ryan") == q.Speaker.Ryan and q.Speaker.Ryan.token_id() == 3061 and q.Language.from_str("en").token_id() == 2050
    assert callable(api.codes_to_tensor) and callable(api.resample_to_24k) and callable(api.auto_device)


def test_committed_bench_line_follows_the_contract():
    """The bench line committed under profiles/ (written by bench.py on the GPU box) carries every field the driver's
    contract names, with the roofline and cpu_baseline objects."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("r1_bench_n1_b8.json", "r2_bench_n1_b8.json"):
        _check_bench_line(json.loads(open(os.path.join(root, "profiles", name)).read().strip().splitlines()[-1]))


def _check_bench_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * 640 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1000.0)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and len(c["sample"]) > 10


def _dumps(tmp_path, codes, pcm, seed=7):
    n = codes.shape[0]
    api.save_codes_binary(str(tmp_path / f"codes_seed{seed}_frames{n}.bin"), codes)
    api.save_audio_binary(str(tmp_path / f"audio_seed{seed}_frames{n}.bin"), pcm)
    with open(tmp_path / f"metadata_seed{seed}_frames{n}.json", "w") as f:
        json.dump({"seed": seed, "num_frames": n}, f)


def test_compare_with_reference_match_and_mismatch(tmp_path):
    """The pin hook (generate_audio.rs:816-931) on synthetic dumps: identical codes + PCM within 1e-3 RMS pin, a single flipped
    code is located by (frame, group), PCM verdicts follow the reference's MATCH / CLOSE / DIFFERENT thresholds."""
    rng = np.random.default_rng(0)
    codes = rng.integers(0, 2048, size=(12, 16)).astype(np.uint32)
    pcm = (0.2 * rng.standard_normal(12 * 1920)).astype(np.float32)
    _dumps(tmp_path, codes, pcm)
    assert api.load_audio_binary(str(tmp_path / "audio_seed7_frames12.bin")).tobytes() == pcm.tobytes()
    lines = []
    rep = cli.compare_with_reference(str(tmp_path), 7, 12, codes, pcm + np.float32(2e-6), out=lines.append)
    assert rep["codes_match"] and rep["n_diff"] == 0 and rep["status"] == "MATCH" and rep["pinned"] is True
    assert any("Codes: MATCH (all 192 values identical)" in l for l in lines) and any("PINNED" in l for l in lines)
    # one flipped code-predictor decision in frame 5, group 9
    bad = codes.copy(); bad[5, 9] ^= 1
    lines = []
    rep = cli.compare_with_reference(str(tmp_path), 7, 12, bad, pcm + np.float32(5e-4), out=lines.append)
    assert rep["codes_match"] is False and rep["n_diff"] == 1 and rep["first_diff"] == (5, 9)
    assert rep["status"] == "CLOSE" and rep["pinned"] is False and 4e-4 < rep["rmse"] < 6e-4
    assert any("Index 89 (frame 5, group 9)" in l for l in lines) and any("NOT pinned" in l for l in lines)
    # different lengths / large PCM error
    rep = cli.compare_with_reference(str(tmp_path), 7, 12, codes[:11], pcm[:11 * 1920] * 0.5, out=lambda *_: None)
    assert rep["codes_match"] is False and rep["status"] == "DIFFERENT" and rep["pinned"] is False


def test_compare_with_reference_missing_files(tmp_path):
    lines = []
    rep = cli.compare_with_reference(str(tmp_path), 1, 4, np.zeros((4, 16), np.uint32), np.zeros(4 * 1920, np.float32), out=lines.append)
    assert rep["codes_found"] is False and rep["audio_found"] is False and rep["pinned"] is None
    assert sum("reference not found" in l for l in lines) == 2
    a = cli.build_parser().parse_args(["--compare", "--reference-dir", "x", "--custom-voice"])
    assert a.compare and a.reference_dir == "x" and a.custom_voice and not a.compare_strict
    assert cli.build_parser().parse_args([]).reference_dir == "test_data/reference_audio"     # generate_audio.rs:73-75


@pytest.mark.gpu
def test_cli_compare_round_trip(tmp_path):
    """--compare end to end: a first run writes the dumps (EOS off, fixed length), a second run with the same flags compares
    against them and pins; a run with another seed is reported as diverging and --compare-strict turns that into exit 1."""
    base = ["--synthetic", "tiny", "--text", "pin me down", "--frames", "9", "--temperature", "0.9", "--output-dir"]
    assert cli.main(base + [str(tmp_path / "ref"), "--seed", "5", "--compare", "--reference-dir", str(tmp_path / "none")]) == 0
    assert os.path.exists(tmp_path / "ref" / "codes_seed5_frames9.bin")          # --compare ran all 9 frames (no EOS)
    assert cli.main(base + [str(tmp_path / "a"), "--seed", "5", "--compare", "--compare-strict", "--reference-dir", str(tmp_path / "ref")]) == 0
    os.rename(tmp_path / "ref" / "codes_seed5_frames9.bin", tmp_path / "ref" / "codes_seed6_frames9.bin")
    os.rename(tmp_path / "ref" / "audio_seed5_frames9.bin", tmp_path / "ref" / "audio_seed6_frames9.bin")
    assert cli.main(base + [str(tmp_path / "b"), "--seed", "6", "--compare", "--compare-strict", "--reference-dir", str(tmp_path / "ref")]) == 1
