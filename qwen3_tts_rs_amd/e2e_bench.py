"""`python -m qwen3_tts_rs_amd.e2e_bench` — the reference's end-to-end benchmark (benches/e2e_bench.rs) with its report
schema (BenchmarkReport / BenchmarkResult / StageBreakdown, e2e_bench.rs:64-115): short / medium / long sentences,
warm-up + timed iterations, wall-clock statistics, RTF, frames per second, optional streaming TTFA, per-stage means."""
import argparse
import json
import statistics
import sys
import time

CORPUS = [
    ("short", "The quick brown fox jumps over the lazy dog near the river bank."),
    ("medium", "In a quiet village nestled between rolling hills and dense forests, there lived an old clockmaker who spent his "
               "days repairing timepieces from centuries past. His workshop, filled with the gentle ticking of a hundred clocks, "
               "was a place where time itself seemed to slow down and the outside world faded into silence."),
    ("long", "The development of artificial intelligence has been one of the most transformative technological advances of the "
             "twenty-first century. From natural language processing to computer vision, machine learning models have achieved "
             "remarkable performance across a wide range of tasks that were once considered the exclusive domain of human "
             "intelligence. Speech synthesis, in particular, has seen dramatic improvements with the introduction of neural "
             "network architectures that can generate high-fidelity audio from text input. These systems learn complex patterns "
             "of prosody, intonation, and rhythm from large datasets of recorded speech, producing output that is increasingly "
             "difficult to distinguish from natural human speech. The implications of this technology extend across many fields, "
             "including accessibility, entertainment, education, and human-computer interaction."),
]


def peak_memory_mb():
    try:
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS:"):
                return float(line.split()[1]) / 1024.0
    except OSError:
        pass
    return None


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="qwen3_tts_rs_amd.e2e_bench")
    ap.add_argument("--device", default="auto")
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--synthetic", choices=["tiny", "0.6b", "1.7b"], default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--json-output", default=None)
    ap.add_argument("--streaming", action="store_true", help="also measure time-to-first-audio via streaming")
    ap.add_argument("--only", default=None, help='only these labels (comma-separated, e.g. "short,medium")')
    ap.add_argument("--max-frames", type=int, default=2048, help="cap per utterance (synthetic weights rarely emit EOS)")
    a = ap.parse_args(argv)
    import qwen3_tts_rs_amd as q
    from qwen3_tts_rs_amd import api
    from qwen3_tts_rs_amd.cli import parse_device
    from qwen3_tts_rs_amd.text import TextTokenizer
    if not a.model_dir and not a.synthetic:
        ap.error("--model-dir or --synthetic is required")
    dev = parse_device(a.device)
    if a.synthetic:
        model = q.Qwen3TTS.from_synthetic({"tiny": q.tiny, "0.6b": q.qwen3_tts_0_6b, "1.7b": q.qwen3_tts_1_7b}[a.synthetic](), device=dev)
        tok = TextTokenizer.from_pretrained(None, allow_stand_in=True)
    else:
        model = q.Qwen3TTS.from_pretrained(a.model_dir, device=dev)
        tok = TextTokenizer.from_pretrained(a.model_dir)
    only = set(a.only.split(",")) if a.only else None
    spf = model.config.samples_per_frame
    results = []
    for label, text in CORPUS:
        if only and label not in only:
            continue
        ids = tok.encode(text)
        opts = q.SynthesisOptions(seed=a.seed, max_length=a.max_frames)

        def run_single():
            utt = q.Utterance(ids, q.Speaker.Ryan, q.Language.English, seed=a.seed)
            t0 = time.perf_counter(); ttfa = None
            if a.streaming:
                ss = api.StreamingSession(model, utt, opts); n = 0
                for c in ss:
                    if ttfa is None:
                        ttfa = (time.perf_counter() - t0) * 1e3
                    n += len(c)
                ss._s.close()
                return (time.perf_counter() - t0) * 1e3, n // spf, n, ttfa, None
            s = model.session([utt], opts)
            audio, timing = s.run(); s.close()
            return (time.perf_counter() - t0) * 1e3, timing.generation_frames, len(audio[0]), None, timing

        for _ in range(a.warmup):
            run_single()
        walls, ttfas, timings, frames, samples = [], [], [], 0, 0
        for _ in range(a.iterations):
            w, frames, samples, t, tm = run_single()
            walls.append(w)
            if t is not None:
                ttfas.append(t)
            if tm is not None:
                timings.append(tm)
        mean = statistics.fmean(walls)
        dur = samples / 24000.0
        res = {"label": label, "text": text, "word_count": len(text.split()), "wall_clock_ms": mean,
               "wall_clock_stddev_ms": statistics.pstdev(walls) if len(walls) > 1 else 0.0,
               "wall_clock_min_ms": min(walls), "wall_clock_max_ms": max(walls), "audio_duration_secs": dur,
               "rtf": (mean / 1e3) / dur if dur > 0 else float("inf"),
               "ttfa_ms": statistics.fmean(ttfas) if ttfas else None,
               "tokens_per_sec": frames / (mean / 1e3) if mean > 0 else 0.0, "frames_generated": frames,
               "peak_memory_mb": peak_memory_mb(),
               "stages": {"prefill_ms": statistics.fmean(t.prefill_ms for t in timings), "generation_ms": statistics.fmean(t.generation_ms for t in timings),
                          "generation_frames": timings[-1].generation_frames, "decode_ms": statistics.fmean(t.decode_ms for t in timings)} if timings else None}
        results.append(res)
        print(f"{label:8s} {res['word_count']:4d} words {len(ids):4d} ids: wall {mean:9.1f} ms  audio {dur:7.2f} s  RTF {res['rtf']:.3f}  "
              f"{res['tokens_per_sec']:.1f} frames/s" + (f"  TTFA {res['ttfa_ms']:.1f} ms" if res["ttfa_ms"] is not None else ""))
    report = {"device": f"hip:{dev} (MI355X)", "model_dir": a.model_dir or f"synthetic:{a.synthetic}", "iterations": a.iterations,
              "tokenizer": tok.kind, "results": results}
    if a.json_output:
        with open(a.json_output, "w") as f:
            json.dump(report, f, indent=2)
        print(f"Results written to {a.json_output}")
    model.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
