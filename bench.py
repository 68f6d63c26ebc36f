#!/usr/bin/env python3
"""bench.py — RTF / acoustic-frames-per-second of the MI355X-native Qwen3-TTS hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under
torch.distributed.run, one rank per GPU. Prints ONE JSON line on rank 0.

A "step" = one non-streaming synthesis (prefill → generate → decode, the reference's
`synthesize_with_timing`, lib.rs:425-501 / benches/e2e_bench.rs:207-262) of this rank's batch of
utterances: synthetic 512-token prompts, CustomVoice prefill, eos_token_id = None and a fixed frame
count so the work is identical across engines, default sampling (temperature 0.9, top-k 50, top-p
0.9, repetition penalty 1.05, seed 42 + utterance index) — SURVEY.md §8(d).
Weak scaling: every GPU gets `--batch` utterances (config[3] of BASELINE.json: 64 utterances over 8
GPUs = 8 per GPU); value = frames of ALL ranks / max-over-ranks time.

Extra objects: "roofline" (frac = the whole captured frame against the HBM roof, measured by this
run; the GEMV family alone — isolated replay, and inside the graph from the committed rocprof table —
in sub-objects that say which is which) and "cpu_baseline" (the
C oracle = port of the reference's candle-CPU F32 path, timed on this host's cores on a bounded
sample of the same workload; rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Figures that cannot be taken from inside the run (rocprofv3 kernel tables, PMC passes) come from files committed under
# profiles/ for THIS round only: an older round's profile is never substituted silently — the field is null instead.
PROFILE_ROUND = "r6"


def committed_rocprof_table(avg_bytes_per_launch, model, batch):
    """GEMV-family launches INSIDE the captured frame graph, from a rocprofv3 --kernel-trace table COMMITTED under profiles/
    (`kernel calls avg_us ...` rows of a B = 8-only run of this command). It is NOT measured by this run — PMC / kernel
    traces cannot be taken from inside the process — and says so: `measured_in_this_run` is False and `source` names the
    file; the figure this run measures itself is roofline.frac (whole frame, in graph). Returned only for the
    configuration the table was taken on (1.7B, 8 rows)."""
    if model != "1.7b" or batch != 8:
        return None
    for name in (f"{PROFILE_ROUND}_rocprof_kernel_stats_bench_b8.txt",):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        calls = us = 0.0
        for line in open(path):      # rows of tools/prof_analyze.py: kernel  WGs thr calls /frame avg_us gap_us ms/frame %busy
            f = line.split()
            if len(f) >= 9 and f[0].startswith("k_gemv"):
                try:
                    calls += float(f[-6]); us += float(f[-6]) * float(f[-4])
                except ValueError:
                    pass
        if calls:
            avg = us / calls
            return {"measured_in_this_run": False, "source": f"profiles/{name}", "launches": int(calls), "avg_launch_us": avg,
                    "gbps": avg_bytes_per_launch / avg / 1e3, "frac": avg_bytes_per_launch / avg / 1e3 / 8000.0}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="1.7b", choices=["1.7b", "0.6b", "tiny"])
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU (one session; up to 64)")
    ap.add_argument("--also-batches", default="64", help="comma-separated extra batch sizes whose frames/s are reported beside the headline (e.g. 32,64)")
    ap.add_argument("--frames", type=int, default=640, help="frames generated per utterance (eos disabled)")
    ap.add_argument("--prompt-tokens", type=int, default=512)
    ap.add_argument("--workload", default="customvoice", choices=["customvoice", "voicedesign4k", "xvector"],
                    help="prefill flavour: CustomVoice (10 positions; BASELINE configs[1-3]), VoiceDesign with a 4096-token instruct "
                         "prompt (config[4]), or Base-model x-vector voice clone")
    ap.add_argument("--sampling", default="default", choices=["default", "greedy"], help="SURVEY §8d cfg B (0.9/50/0.9/1.05) or cfg A (greedy)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=96, help="frames of the CPU-baseline sample (~12 s of oracle time)")
    ap.add_argument("--cpu-frames-single", type=int, default=8, help="frames of the single-thread CPU-baseline sample (~8 s)")
    ap.add_argument("--profile-frames", type=int, default=6)
    ap.add_argument("--ttfa-reps", type=int, default=5)
    ap.add_argument("--headline-only", action="store_true", help="only the timed steps: no roofline replays, latency / TTFA, other batches, other configurations, "
                    "EOS mix or CPU baseline (the command the committed rocprofv3 kernel table is taken from)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE.json side configurations (greedy, x-vector, 4k-token VoiceDesign, 0.6B single utterance)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = True; args.no_other_configs = True; args.also_batches = ""; args.ttfa_reps = 0

    import numpy as np
    import torch
    same_gpu = os.environ.get("Q3_DP_TEST_SAME_GPU") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks ourselves (one process per GPU), never a silent world = 1
        n_vis = torch.cuda.device_count()
        if n_vis < args.gpus and not same_gpu:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {n_vis} GPU(s) are visible")
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execvp(sys.executable, cmd)
    import qwen3_tts_rs_amd as q
    from qwen3_tts_rs_amd import dp, synth

    rank, local_rank, world = dp.env_rank()
    if world != args.gpus and world > 1:
        args.gpus = world
    if world > 1 and not same_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    # Q3_DP_TEST_SAME_GPU=1 (single-GPU boxes only): every rank on cuda:0 over gloo — exercises the whole multi-rank
    # flow (arena broadcast, finalize on non-root ranks, barriers, max-over-ranks timing) where RCCL cannot run
    # (it refuses two ranks on one device). The numbers of such a run mean nothing.
    dev = 0 if same_gpu else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        dp.init("gloo" if same_gpu else "nccl")

    cfg = {"1.7b": q.qwen3_tts_1_7b, "0.6b": q.qwen3_tts_0_6b, "tiny": q.tiny}[args.model]()

    # ---- weights: rank 0 builds the synthetic checkpoint; one RCCL broadcast of the arena ----
    t_load = time.time()
    oracle_sink = []
    if rank == 0:
        model = q.Qwen3TTS.from_synthetic(cfg, device=dev, seed=synth.DEFAULT_SEED)
    else:
        model = q.Qwen3TTS(cfg, device=dev)
    bcast_bytes = 0
    if world > 1:
        t_b = time.time()
        bcast_bytes = dp.broadcast_arena(model, dev)
        bcast_s = time.time() - t_b
        if rank != 0:
            model.mark_loaded(); model.finalize()
    else:
        bcast_s = 0.0
    load_s = time.time() - t_load

    # ---- workload ----
    from qwen3_tts_rs_amd.synth import synthetic_prompt
    B = args.batch
    n_total = B * world
    my_idx = dp.shard_indices(n_total, rank, world)
    def make_utt(i):
        if args.workload == "voicedesign4k":
            return q.Utterance(synthetic_prompt(args.prompt_tokens, i), language=q.Language.English,
                               instruct_ids=synthetic_prompt(4096, 5000 + i), seed=42 + i)
        if args.workload == "xvector":
            xv = np.random.default_rng(7000 + i).standard_normal(cfg.hidden).astype(np.float32)
            return q.Utterance(synthetic_prompt(args.prompt_tokens, i), language=q.Language.English, xvector=xv, seed=42 + i)
        return q.Utterance(synthetic_prompt(args.prompt_tokens, i), q.Speaker.Ryan, q.Language.English, seed=42 + i)
    utts = [make_utt(i) for i in my_idx]
    samp = dict(temperature=0.0) if args.sampling == "greedy" else {}
    opts = q.SynthesisOptions(max_length=args.frames, eos_token_id=None, seed=42, **samp)
    use_graph = not args.no_graph

    submit_path = [0, 0, 0, 0]    # (path, packets per frame, packets without acquire / release fence) of the timed sessions: q3_session_submit_info / _fences
    phase_ms = []       # (create, run, close) wall per step: the timed step is all three
    # the samples of every utterance land in host memory inside the timed step, as `synthesize` returns them
    # (lib.rs:718-784): one pinned buffer per row, reused step after step (allocated once, outside the timing, like a
    # server's output ring)
    pcm_cap = args.frames * cfg.samples_per_frame
    pinned = {}
    def pcm_bufs(n):
        if n not in pinned:
            pinned[n] = torch.empty((n, pcm_cap), dtype=torch.float32).pin_memory()
        return [(pinned[n][i].data_ptr(), pcm_cap) for i in range(n)]
    pcm_out = pcm_bufs(len(utts))

    def one_step():
        ta = time.perf_counter()
        s = model.session(utts, opts)
        tb = time.perf_counter()
        try:
            return s.run_timing_only(use_graph=use_graph, pcm_out=pcm_out)
        finally:
            tc = time.perf_counter()
            try:
                submit_path[:] = list(s.submit_info()) + list(s.submit_fences())
            except Exception:
                pass
            s.close()
            phase_ms.append([(tb - ta) * 1e3, (tc - tb) * 1e3, (time.perf_counter() - tc) * 1e3])

    for _ in range(args.warmup):
        one_step()
    dp.barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    timings, step_wall = [], []
    for _ in range(args.steps):
        ts = time.perf_counter()
        timings.append(one_step())
        step_wall.append((time.perf_counter() - ts) * 1000.0)
    torch.cuda.synchronize(dev); dp.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    elapsed = dp.max_over_ranks(elapsed, device=f"cuda:{dev}" if world > 1 else None)
    frames_rank = sum(t.generation_frames for t in timings)
    frames_total = dp.sum_over_ranks(float(frames_rank), device=f"cuda:{dev}" if world > 1 else None)

    # what the collective library actually saw (so that a SCALE record can be checked from the JSON alone)
    rccl = {"world": 1, "backend": None, "ranks_seen": [[0, dev]]}
    if world > 1:
        import torch.distributed as dist
        mine = torch.tensor([rank, dev], dtype=torch.int64, device=f"cuda:{dev}" if not same_gpu else "cpu")
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        rccl = {"world": dist.get_world_size(), "backend": dist.get_backend(), "ranks_seen": [[int(t[0]), int(t[1])] for t in seen],
                "same_gpu_test_mode": same_gpu}
        # per-rank view (a straggler or a rank that ran fewer frames is visible from the line alone)
        mine_f = torch.tensor([float(rank), float(frames_rank), float(elapsed_local)], dtype=torch.float64, device=f"cuda:{dev}" if not same_gpu else "cpu")
        got = [torch.zeros_like(mine_f) for _ in range(world)]
        dist.all_gather(got, mine_f)
        rccl["per_rank"] = [{"rank": int(t[0]), "frames": int(t[1]), "seconds": float(t[2]), "frames_per_s": float(t[1]) / float(t[2])} for t in got]
        # A multi-GPU line is only valid when it really ran one rank per device over RCCL: every rank checks the same gathered
        # list, so all of them leave together (no rank is left waiting in a collective).
        if not same_gpu:
            devices = {d for _, d in rccl["ranks_seen"]}
            ranks = {r for r, _ in rccl["ranks_seen"]}
            problem = None
            if rccl["backend"] != "nccl": problem = f"backend is {rccl['backend']!r}, not 'nccl' (= RCCL on ROCm)"
            elif rccl["world"] != args.gpus or len(ranks) != args.gpus: problem = f"{len(ranks)} distinct rank(s) of world {rccl['world']} for --gpus {args.gpus}"
            elif len(devices) != args.gpus: problem = f"{len(devices)} distinct device(s) for {args.gpus} ranks: {sorted(devices)}"
            if problem:
                dist.destroy_process_group()
                raise SystemExit(f"bench.py: invalid multi-GPU run — {problem}; rccl = {json.dumps(rccl)}")

    def finish():
        # every rank leaves through the same door: ranks != 0 wait here until rank 0 has printed its line (its roofline /
        # latency extras take tens of seconds; the collective timeout is minutes), then the group is torn down
        if world > 1:
            import torch.distributed as dist
            dp.barrier()
            dist.destroy_process_group()

    if rank != 0:
        finish()
        return

    ms_per_step = elapsed / args.steps * 1000.0
    fps = frames_total / elapsed
    audio_s_per_utt = args.frames * cfg.samples_per_frame / 24000.0
    rtf_job = (elapsed / args.steps) / (audio_s_per_utt * n_total)      # wall per second of audio, whole job
    rtf_utt = (elapsed / args.steps) / audio_s_per_utt                  # latency view: each utterance's RTF
    stage = {"prefill_ms": float(np.mean([t.prefill_ms for t in timings])),
             "generation_ms": float(np.mean([t.generation_ms for t in timings])),
             "decode_ms": float(np.mean([t.decode_ms for t in timings]))}

    # (measured right behind the timed steps; round 6 saw 2496-2578 frames/s from this leg box to box, 2558-2587 from the same loop
    #  alone in a process, tools/dev/eos_mix_ab.py — profiles/r6_eos_mix_staged_ab.txt)
    # ---- utterances that END AT DIFFERENT FRAMES (what EOS does to a real batch): 4 x B requests with lengths drawn from
    # 100 .. frames go through B rows, (a) continuously — the native batcher (q3_batcher_*) refills a row that ends at the next
    # 8-frame step (q3_session_replace) — and (b) as lockstep sessions of B, each running until its longest row is done. Useful frames / wall of the generation
    # loop (prefills of the swapped-in requests included; no vocoder on either side). ----
    eos_mix = None
    if world == 1 and not args.no_other_configs and not args.headline_only and args.workload == "customvoice":
        try:
            rng = np.random.default_rng(2026)
            n_req = 4 * B
            lens = [int(x) for x in rng.integers(min(100, args.frames), args.frames + 1, size=n_req)]
            mix = []
            for i, L in enumerate(lens):
                u = make_utt(i); u.max_length = L
                mix.append(u)
            def batcher_run(reqs, poll, want_pcm=False):            # the native serving loop (q3_batcher): queue -> rows, refilled every `poll` frames
                bt = q.Batcher(model, slots=B, frame_budget=args.frames, prompt_budget=0, options=opts)
                try:
                    ta = time.perf_counter()
                    tickets = [bt.submit(u, want_pcm=want_pcm) for u in reqs]
                    steady = None                    # (frames, seconds) at the moment the queue ran dry: every row busy until then
                    while True:
                        running, queued, _ = bt.step(poll, use_graph)
                        if queued == 0 and steady is None:
                            steady = (sum(bt.poll(t)[1] for t in tickets), time.perf_counter() - ta)
                        if running == 0 and queued == 0:
                            break
                    frames = sum(int(bt.fetch(t)[0].shape[0]) for t in tickets)      # (a finished row's vocoder runs on the batcher's worker: fetch waits for the last samples)
                    wall = time.perf_counter() - ta
                    return frames, wall, steady
                finally:
                    bt.close()
            batcher_run(mix[:B + 2], 8)              # warm (graph, side-session shapes)
            fr_c, wall_c, steady = batcher_run(mix, 8)
            fr_p, wall_p, steady_p = batcher_run(mix, 8, want_pcm=True)      # every finished row vocoded (decode worker)
            wall_l = 0.0; fr_l = 0
            for k in range(0, n_req, B):
                sl = model.session(mix[k:k + B], opts)
                ta = time.perf_counter(); sl.prefill(); sl.generate(args.frames, use_graph=use_graph); wall_l += time.perf_counter() - ta
                fr_l += sum(sl.frames(b)[0] for b in range(len(mix[k:k + B]))); sl.close()
            eos_mix = {"requests": n_req, "rows": B, "lengths": f"uniform {min(100, args.frames)}..{args.frames} frames (seed 2026), mean {float(np.mean(lens)):.0f}",
                       "continuous_frames_per_s": fr_c / wall_c, "continuous_steady_frames_per_s": steady[0] / steady[1],
                       "continuous_with_vocoder_frames_per_s": fr_p / wall_p, "continuous_with_vocoder_steady_frames_per_s": steady_p[0] / steady_p[1],
                       "lockstep_frames_per_s": fr_l / wall_l, "frames": fr_c,
                       "what": "generation loop only (q3_batcher with 8-frame steps; session opening and the prefill of swapped-in requests included, no vocoder except in the `with_vocoder` figures, where every finished row is decoded to PCM (on the batcher's decode worker, beside the frames) and the wall clock stops after the last fetch; `steady` = until the queue ran dry, i.e. without the drain of the last rows); lockstep = sessions of `rows` requests "
                               "each running until its longest row ends"}
        except Exception as e:
            eos_mix = {"error": str(e)}

    # ---- roofline of the dominant kernel: the bf16-weight MFMA GEMV family (every projection of the frame) ----
    # The launch inventory is the ENGINE's: a profiled session runs a few frames and reports every distinct GEMV launch
    # (M, N, K, epilogue, fused input norm, tiling) with its count per frame. Each shape is then replayed
    # from a hipGraph over HBM-resident weight copies and timed with HIP events on the launch stream (q3_bench_linear, mean
    # of 5 replays); achieved = Σ algorithmic weight bytes of one frame's GEMV launches ÷ Σ their launch times. The same
    # profiled frames also give the in-situ figure (event pairs around every GEMV launch of real frames, eager launches).
    from qwen3_tts_rs_amd.api import bench_linear
    pf = max(2, args.profile_frames)
    sp = model.session(utts, q.SynthesisOptions(max_length=pf + 2, eos_token_id=None, seed=42, **samp))
    wbytes, kvbytes = sp.frame_bytes((10 if args.workload != "voicedesign4k" else 4105) + args.frames // 2)
    insitu_ms = insitu_bytes = 0.0; insitu_n = 0; shapes = []
    if not args.headline_only:
        sp.prefill(); sp.generate(1, use_graph=False)             # first frame outside the inventory (lazy initialisation)
        sp.set_profile(True); sp.profile_shapes(reset=True); sp.profile_read(reset=True)
        sp.generate(pf, use_graph=False)
        insitu_ms, insitu_bytes, insitu_n = sp.profile_read(reset=True)
        shapes = sp.profile_shapes(reset=True)
    sp.close()
    tot_bytes = tot_us = 0.0; launches = 0; per_shape = {}
    EPI = {0: "none", 1: "resid", 2: "silu", 3: "swiglu"}
    for (Mr, N, K, epi, rms, _reserved, tiled, count) in shapes:
        assert count % pf == 0, (Mr, N, K, count, pf)
        cnt = count // pf
        nb = N * K * 2 * (2 if epi == 3 else 1)
        try:
            us = bench_linear(Mr, N, K, epi, rms, tiled=tiled, device=dev)
        except Exception as e:          # a shape the replay harness refuses (tiny test configurations): leave it out of the sum
            per_shape[f"M={Mr} N={N} K={K} {EPI[epi]} norm={rms} tile={tiled}"] = {"error": str(e), "launches_per_frame": cnt}
            continue
        per_shape[f"M={Mr} N={N} K={K} {EPI[epi]} norm={rms} tile={ {1: '16', 2: '4', 3: '16 split-K2'}.get(tiled, tiled) }"] = \
            {"us": us, "gbps": nb / us / 1e3, "launches_per_frame": cnt}
        tot_bytes += nb * cnt; tot_us += us * cnt; launches += cnt
    if not launches:
        tot_bytes, tot_us, launches = 1.0, 1.0, 1
    achieved = tot_bytes / tot_us / 1e3     # GB/s
    # HBM traffic per launch from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # passes, FETCH_SIZE x2 gfx950 correction; tools/pmc_collect.sh) — PMC counters cannot be read from inside the run
    traffic = None; pmc_path = None
    cand = f"{PROFILE_ROUND}_pmc_gemv_M{min(B, 16)}.json"          # this round's profile or nothing (traffic: null)
    if os.path.exists(os.path.join(ROOT, "profiles", cand)):
        pmc_path = os.path.join(ROOT, "profiles", cand)
    if args.model == "1.7b" and pmc_path:
        pmc = json.load(open(pmc_path)).get("shapes", {})
        by_dims = {(v["N"], v["K"], v["epi"]): v["fetch_bytes_corrected"] + v["write_bytes"] for v in pmc.values() if "N" in v}
        tsum = tcnt = 0.0
        for (Mr, N, K, epi, rms, _reserved, tiled, count) in shapes:      # weighted by the engine's launches per frame
            if (N, K, epi) in by_dims:
                tsum += by_dims[(N, K, epi)] * (count // pf); tcnt += count // pf
        if tcnt >= 0.9 * launches:
            traffic = tsum / tcnt
        elif pmc:                                                       # older profile without dims: plain mean over its shapes
            traffic = float(np.mean([v["fetch_bytes_corrected"] + v["write_bytes"] for v in pmc.values()]))
    # Which number is which:
    #   frac / achieved  = the WHOLE FRAME inside the captured graph, measured by this run: (weight bytes of every launch of a
    #                      frame + the KV bytes its attention reads at the mean context) / (generation wall time / frames);
    #   step_frac        = the same bytes over the whole timed step (prefill, vocoder, PCM copy-out, session open / close);
    #   isolated         = the GEMV family with each shape replayed ALONE from a hipGraph (no neighbours: the ceiling of the
    #                      kernels, not what the frame gets);
    #   gemv_in_graph    = the GEMV family inside the frame, from the COMMITTED rocprofv3 table (not measured by this run).
    frame_s = stage["generation_ms"] / 1000.0 / args.frames
    frame_gbps = (wbytes + kvbytes) / frame_s / 1e9
    step_gbps = (wbytes + kvbytes) * args.frames / (ms_per_step / 1000.0) / 1e9
    roofline = {"bound": "hbm", "achieved": frame_gbps, "peak": 8000.0, "unit": "GB/s", "frac": frame_gbps / 8000.0,
                "frame_frac": frame_gbps / 8000.0, "step_frac": step_gbps / 8000.0,
                "traffic": traffic, "traffic_source": os.path.basename(pmc_path) if (traffic and pmc_path) else None,
                "kernel": "the captured frame (one hipGraph replay = talker step + 15 code-predictor passes + sampler); dominant family: "
                          "k_gemv_mfma / k_gemv_sk2 / k_gemv_mfma4 / k_gemv_lds (bf16-weight MFMA GEMV, M = batch)",
                "timing": "frac: generation wall time of the timed steps / frames (graph replays on the session stream, host clock around "
                          "the stream-synchronised loop); traffic: per GEMV launch, committed PMC passes",
                "frame_weight_bytes": wbytes, "frame_kv_bytes": kvbytes, "frame_ms": frame_s * 1e3,
                "isolated": {"what": "GEMV family, each shape replayed alone from a hipGraph over HBM-resident weight copies (launch inventory from "
                                     "the engine's profiled frames), HIP events on the launch stream, mean of 5 replays x 200 launches",
                             "gbps": achieved, "frac": achieved / 8000.0, "launches_per_frame": launches, "avg_launch_us": tot_us / launches,
                             "avg_bytes_per_launch": tot_bytes / launches, "gemv_us_per_frame": tot_us, "per_shape": per_shape},
                "eager_in_situ": {"what": "HIP event pairs around every GEMV launch of real frames (EAGER launches in a profiling session): inflated by "
                                          "the event records and the host launch path — a lower bound, not the figure to trust",
                                  "launches_per_frame": insitu_n / pf, "avg_launch_us": insitu_ms * 1e3 / max(insitu_n, 1),
                                  "gbps": insitu_bytes / max(insitu_ms, 1e-9) / 1e6, "frac": insitu_bytes / max(insitu_ms, 1e-9) / 1e6 / 8000.0},
                "gemv_in_graph": committed_rocprof_table(tot_bytes / launches, args.model, B)}

    # ---- single-utterance latency + streaming TTFA (config[2]) ----
    lat = {}
    try:
        if args.headline_only:
            raise RuntimeError("skipped (--headline-only)")
        u1 = [q.Utterance(synthetic_prompt(args.prompt_tokens, 0), seed=42)]
        f1 = min(args.frames, 160)
        s1 = model.session(u1, q.SynthesisOptions(max_length=f1, eos_token_id=None, seed=42))
        t1 = s1.run_timing_only(use_graph=use_graph); s1.close()
        s1 = model.session(u1, q.SynthesisOptions(max_length=f1, eos_token_id=None, seed=42))
        t1 = s1.run_timing_only(use_graph=use_graph); s1.close()
        wall = (t1.prefill_ms + t1.generation_ms + t1.decode_ms) / 1000.0
        lat = {"b1_frames": f1, "b1_frames_per_s": f1 / wall, "b1_rtf": wall / (f1 * 0.08), "b1_prefill_ms": t1.prefill_ms,
               "b1_ms_per_frame": t1.generation_ms / f1, "b1_decode_ms": t1.decode_ms}
        ttfa = []
        for _ in range(args.ttfa_reps):
            ta = time.perf_counter()
            ss = model.synthesize_streaming(u1[0].text_ids, q.Speaker.Ryan, q.Language.English,
                                            q.SynthesisOptions(max_length=30, eos_token_id=None, seed=42, chunk_frames=10))
            ss.next_chunk()
            ttfa.append((time.perf_counter() - ta) * 1000.0)
            ss._s.close()
        lat["ttfa_ms_p50"] = float(np.median(ttfa))
    except Exception as e:   # latency extras must never kill the headline line
        lat["error"] = str(e)

    # ---- wider sessions on the same GPU (serving view): frames/s of one step at other batch sizes ----
    wide = {}
    for bb in [int(x) for x in args.also_batches.split(",") if x.strip()]:
        try:
            uw = [make_utt(i) for i in range(bb)]
            for rep in range(2):            # first pass warms the session-shape cache
                pw = pcm_bufs(bb)
                sw = model.session(uw, opts); tw0 = time.perf_counter(); tw = sw.run_timing_only(use_graph=use_graph, pcm_out=pw); tw1 = time.perf_counter(); sw.close()
            wide[str(bb)] = {"frames_per_s": tw.generation_frames / (tw1 - tw0), "ms_per_frame": tw.generation_ms / args.frames,
                             "stage_ms": {"prefill_ms": tw.prefill_ms, "generation_ms": tw.generation_ms, "decode_ms": tw.decode_ms}}
        except Exception as e:
            wide[str(bb)] = {"error": str(e)}

    # ---- BASELINE.json's other configurations, same step definition, one warm + one timed step each (N = 1 only) ----
    others = {}
    if world == 1 and not args.no_other_configs and args.workload == "customvoice" and args.sampling == "default":
        def timed(mdl, uu, oo, reps=1, kv_bf16=False):
            sw = mdl.session(uu, oo, kv_bf16=kv_bf16); sw.run_timing_only(use_graph=use_graph); sw.close()           # warm (graph capture, session-shape cache)
            best = None; po = pcm_bufs(len(uu))
            for _ in range(reps):
                sw = mdl.session(uu, oo, kv_bf16=kv_bf16); ta = time.perf_counter(); tt = sw.run_timing_only(use_graph=use_graph, pcm_out=po); wall = time.perf_counter() - ta; sw.close()
                if best is None or wall < best[0]:
                    best = (wall, tt)
            wall, tt = best
            return {"frames_per_s": tt.generation_frames / wall, "rtf_per_utterance": wall / (args.frames * 0.08), "ms_per_frame": tt.generation_ms / args.frames,
                    "stage_ms": {"prefill_ms": tt.prefill_ms, "generation_ms": tt.generation_ms, "decode_ms": tt.decode_ms}, "utterances": len(uu)}
        def guarded(name, fn):
            try:
                others[name] = fn()
            except Exception as e:      # side configurations must never kill the headline line
                others[name] = {"error": str(e)}
        saved = args.workload
        guarded(f"{args.model}_greedy_b{B}", lambda: timed(model, utts, q.SynthesisOptions(max_length=args.frames, eos_token_id=None, seed=42, temperature=0.0)))
        args.workload = "xvector"
        guarded(f"{args.model}_xvector_b{B}", lambda: timed(model, [make_utt(i) for i in range(B)], opts))
        args.workload = "voicedesign4k"
        guarded(f"{args.model}_voicedesign4k_b1", lambda: timed(model, [make_utt(0)], opts))
        args.workload = saved
        # the reference GPU path's cache dtype as an opt-in session mode (q3_session_set_kv_dtype): NOT the headline — results are
        # no longer bit-comparable with the F32 oracle; half the K/V bytes per frame
        guarded(f"{args.model}_bf16kv_b{B}", lambda: timed(model, utts, opts, kv_bf16=True))
        guarded(f"{args.model}_bf16kv_b64", lambda: timed(model, [make_utt(i) for i in range(64)], opts, kv_bf16=True))
        # the vocoder's convs on two bf16 planes per operand instead of three (q3_model_set_codec_planes(2)): an opt-in mode, NOT the
        # headline — token ids are the same bits, the PCM is within 1e-4 RMS of the CPU path instead of 2.5e-5 (tolerance 1e-3)
        def planes2(uu, kv_bf16=False):
            model.set_codec_planes(2)
            try:
                return timed(model, uu, opts, kv_bf16=kv_bf16)
            finally:
                model.set_codec_planes(3)
        guarded(f"{args.model}_codec2p_b{B}", lambda: planes2(utts))
        guarded(f"{args.model}_codec2p_b64", lambda: planes2([make_utt(i) for i in range(64)]))
        # both opt-in modes together (bf16 K/V pages + two-plane vocoder): what a throughput-first deployment would run
        guarded(f"{args.model}_bf16kv_codec2p_b64", lambda: planes2([make_utt(i) for i in range(64)], kv_bf16=True))
        if args.model == "1.7b":
            def small():
                m06 = q.Qwen3TTS.from_synthetic(q.qwen3_tts_0_6b(), device=dev, seed=synth.DEFAULT_SEED)
                try:
                    return timed(m06, [q.Utterance(synthetic_prompt(args.prompt_tokens, 0), q.Speaker.Ryan, q.Language.English, seed=42)], opts)
                finally:
                    m06.close()
            guarded("0.6b_customvoice_b1", small)

    # ---- CPU baseline: the oracle (port of the candle-CPU F32 path) on this host, bounded sample ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))      # the oracle wrapper is test infrastructure: only this baseline leg touches it
            import oracle as O
            from common import oracle_model
            t_o = time.time()
            om = oracle_model(cfg, synth.DEFAULT_SEED, which=3)
            o_load = time.time() - t_o
            try:
                O.olib.q3o_get_threads.restype = ctypes.c_int
                ncores = int(O.olib.q3o_get_threads())      # threads the oracle's parallel regions actually use (min(cores, 32))
            except AttributeError:
                ncores = min(os.cpu_count() or 1, 32)
            utt0 = q.Utterance(synthetic_prompt(args.prompt_tokens, 0), seed=42)
            oo = q.SynthesisOptions(max_length=args.cpu_frames, eos_token_id=None, seed=42)
            tc = time.perf_counter()
            osess = O.OracleSession(om, utt0, oo)
            codes = osess.generate()
            pcm = om.decode(codes)
            cpu_wall = time.perf_counter() - tc
            osess.close()
            # single thread (deterministic, one core): a few frames of the same utterance
            single = None
            try:
                O.olib.q3o_set_threads(1)
                o1 = q.SynthesisOptions(max_length=args.cpu_frames_single, eos_token_id=None, seed=42)
                t1c = time.perf_counter()
                os1 = O.OracleSession(om, utt0, o1); c1 = os1.generate(); p1 = om.decode(c1)
                w1 = time.perf_counter() - t1c
                os1.close()
                single = {"value": len(c1) / w1, "unit": "frames/s", "cores": 1, "rtf": w1 / (len(c1) * 0.08),
                          "sample": f"prefill + {len(c1)} frames + vocoder, {w1:.1f}s wall"}
            except Exception as e1:
                single = {"value": None, "sample": f"failed: {e1}"}
            finally:
                O.olib.q3o_set_threads(ncores)
            om.close()
            cpu = {"value": len(codes) / cpu_wall, "unit": "frames/s", "cores": ncores, "kind": "port",
                   "rtf": cpu_wall / (len(codes) * 0.08), "single_thread": single,
                   "sample": f"1 utterance, {args.prompt_tokens}-token prompt: prefill + {len(codes)} frames + vocoder "
                             f"({cpu_wall:.1f}s wall, oracle load {o_load:.0f}s not counted); reference-published CPU: "
                             f"2.3/2.1/1.9 frames/s, RTF 5.39-6.48 on 20 Arm cores (docs/BENCHMARKS.md:111-115)"}
        except Exception as e:
            cpu = {"value": None, "unit": "frames/s", "cores": min(os.cpu_count() or 1, 32), "kind": "port", "sample": f"failed: {e}"}

    out = {
        "metric": "acoustic_frames_per_sec", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Qwen3-TTS-{args.model} synthetic weights; {B} utterances/GPU x {world} GPU(s), "
                               f"{args.prompt_tokens}-token prompts, {args.workload} prefill, {args.frames} frames each "
                               f"(eos off), {args.sampling} sampling, non-streaming prefill+generate+decode",
                   "utterances_per_gpu": B, "frames_per_utterance": args.frames, "parallelism": f"dp{world}",
                   "weights": "bf16", "activations_kv": "f32",
                   "kv": "f32, in place (2x the bytes of the reference GPU path's bf16 cache, kv_cache.rs:234-310: the parity contract is the CPU F32 path)",
                   "hip_graph": use_graph,
                   "frame_submit": {0: "eager launches", 1: "hipGraphLaunch", 2: "own AQL queue, HIP's fences on every packet",
                                    3: "own AQL queue, no fences (probe: invalid)",
                                    4: "captured frame replayed on the library's own AQL queue; boundaries between write-through kernels without agent-scope fences"}.get(submit_path[0], str(submit_path[0])),
                   "frame_packets": submit_path[1], "frame_packets_without_acquire_release_fence": submit_path[2:4], "prefill": args.workload, "sampling": args.sampling,
                   "pcm_copy_out": True},      # every utterance's samples are copied to (pinned) host memory inside the timed step
        "rtf": rtf_job, "rtf_per_utterance": rtf_utt, "stage_ms": stage, "step_wall_ms": step_wall, "step_phase_ms_create_run_close": phase_ms[-args.steps:], "latency": lat,
        "weights_load_s": load_s, "weight_broadcast": {"bytes": bcast_bytes, "seconds": bcast_s,
                                                        "gbps": (bcast_bytes / bcast_s / 1e9) if bcast_s > 0 else None,
                                                        "expected": "one RCCL broadcast of the 4.4 GB arena from rank 0: ~30 ms per-link bound on xGMI (~150 GB/s), "
                                                                    "more on the first call (communicator set-up); steady state has no collective (DESIGN 6)"},
        "roofline": roofline, "cpu_baseline": cpu, "other_batches": wide, "other_configs": others, "eos_mix": eos_mix, "rccl": rccl,
    }
    print(json.dumps(out), flush=True)
    finish()


if __name__ == "__main__":
    main()
