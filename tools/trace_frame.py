"""Gantt table of ONE frame of the captured frame graph at 10 ns resolution (development aid; needs the -DQ3_TRACE build:
bash tools/trace_build.sh here, then ON the GPU box  python tools/trace_frame.py [model] [B] [frames] [prompt]).

Every instrumented kernel (the GEMV family, the decode attentions, the merge) stores s_memrealtime stamps per workgroup:
slot 0 entry, 1 inputs landed / B operand ready, 2 main loop done (all loads landed), 3 results stored (issued),
4 stores acknowledged. Printed per node: gap = first entry - last exit of the previous instrumented node (the launch
boundary as the GPU sees it), skew = last entry - first entry (dispatch spread over the workgroups), then the slot times
relative to the node's first entry as median / max over its workgroups, and the node's total. Un-instrumented nodes
(sampler, frame embed, rmsnorm, first2) show up as larger gaps."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("Q3TTS_LIB", os.path.join(ROOT, "qwen3_tts_rs_amd", "libq3tts_trace.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import _lib
from common import synthetic_prompt

model = sys.argv[1] if len(sys.argv) > 1 else "1.7b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 64
prompt = int(sys.argv[4]) if len(sys.argv) > 4 else 512
full = "--full" in sys.argv
SLOTS, WGS, DWGS, WAVES = 8, 512, 16, 16
NODE = SLOTS * (WGS + DWGS * WAVES)
lib = _lib.lib
cfg = {"1.7b": q.qwen3_tts_1_7b, "0.6b": q.qwen3_tts_0_6b, "tiny": q.tiny}[model]()
m = q.Qwen3TTS.from_synthetic(cfg)
utts = [q.Utterance(synthetic_prompt(prompt, i), seed=42 + i) for i in range(B)]
s = m.session(utts, q.SynthesisOptions(max_length=frames, eos_token_id=None, seed=42))
s.prefill()
_lib.check(lib.q3_debug_trace_enable(s._h, 800))
s.generate(frames, use_graph=True)
n = ctypes.c_int()
_lib.check(lib.q3_debug_trace_read(s._h, None, None, 0, ctypes.byref(n)))
N = n.value
raw = np.zeros((N, NODE), np.uint64); meta = np.zeros((N, 7), np.int32)
_lib.check(lib.q3_debug_trace_read(s._h, raw.ctypes.data_as(ctypes.c_void_p), meta.ctypes.data_as(ctypes.c_void_p), N, ctypes.byref(n)))
s.close()
st = raw[:, :WGS * SLOTS].reshape(N, WGS, SLOTS)
det = raw[:, WGS * SLOTS:].reshape(N, DWGS, WAVES, SLOTS)     # every wave of the first DWGS workgroups

EPI = {0: "none", 1: "resid", 2: "silu", 3: "swiglu"}
def name(mt):
    k = mt[0]
    if k == 0:
        return f"gemv M{mt[1]} N{mt[2]} K{mt[3]} {EPI[mt[4]]}{' rms' if mt[5] else ''} t{16 if mt[6] == 1 else 4}"
    return {1: "attn_cp", 2: "attn_fused", 3: "attn_merge"}[k] + f" B{mt[1]} splits{mt[3]} pos{mt[4]}{' fold' if mt[5] else ''}"

rows = []; prev_out = None
for i in range(N):
    v = st[i]; ok = v[:, 0] != 0
    if not ok.any():
        rows.append(None); continue
    t = v[ok].astype(np.int64)
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01                      # us
    rel[t == 0] = np.nan
    out = np.nanmax(rel[:, 4])
    r = {"i": i, "name": name(meta[i]), "wgs": int(ok.sum()), "gap": (t0 - prev_out) * 0.01 if prev_out is not None else np.nan,
         "skew": float(rel[:, 0].max()), "tot": float(out)}
    for k in (1, 2, 3, 4, 7):
        col = rel[:, k]
        r[f"s{k}m"] = float(np.nanmedian(col)) if np.isfinite(col).any() else np.nan
        r[f"s{k}x"] = float(np.nanmax(col)) if np.isfinite(col).any() else np.nan
    # per-wave detail: when did each wave finish its main loop (slot 2) / its partial writes (6), when did the barrier open (5),
    # all relative to the node's first entry; "tail" = store issued (3) - barrier opened (5) on the storing waves
    d = det[i].astype(np.int64); dok = d[:, :, 0] != 0
    if dok.any():
        dr = (d - t0) * 0.01; dr[d == 0] = np.nan
        l2 = dr[:, :, 2][dok]; r["w2"] = (float(np.nanmin(l2)), float(np.nanmedian(l2)), float(np.nanmax(l2)))
        e0 = dr[:, :, 0][dok]; r["w0x"] = float(np.nanmax(e0))
        if meta[i][0] == 1:      # attn_cp: slots 5 / 6 hold clock64() counters (the shader-clock line below), not time stamps
            r["w5"] = r["w6x"] = r["tail"] = np.nan
        else:
            b5 = dr[:, :, 5][dok]; r["w5"] = float(np.nanmedian(b5)) if np.isfinite(b5).any() else np.nan
            w6 = dr[:, :, 6][dok]; r["w6x"] = float(np.nanmax(w6)) if np.isfinite(w6).any() else np.nan
            s3 = dr[:, 0, 3]; s5 = dr[:, 0, 5]
            r["tail"] = float(np.nanmedian(s3 - s5)) if np.isfinite(s3 - s5).any() else np.nan
    prev_out = t0 + int(round(out * 100))
    rows.append(r)

hdr = f"{'node':>4} {'kernel':44} {'WGs':>4} {'gap':>6} {'skew':>5} | {'in med/max':>11} {'loop med/max':>12} {'st med/max':>11} {'ack med/max':>11} | {'total':>6}"
def line(r):
    return (f"{r['i']:4d} {r['name']:44} {r['wgs']:4d} {r['gap']:6.2f} {r['skew']:5.2f} | {r['s1m']:5.2f}/{r['s1x']:5.2f} {r['s2m']:6.2f}/{r['s2x']:5.2f} "
            f"{r['s3m']:5.2f}/{r['s3x']:5.2f} {r['s4m']:5.2f}/{r['s4x']:5.2f} | {r['tot']:6.2f}")
print(f"# {model} B={B} frame {frames} (context {prompt and 10}+{frames}), {N} instrumented nodes; times in us")
if full:
    print(hdr)
    for r in rows:
        if r: print(line(r))
# aggregate by kernel name
agg = {}
for r in rows:
    if r: agg.setdefault(r["name"], []).append(r)
print("\n# mean per kernel shape (count = nodes per frame)")
print(f"{'kernel':44} {'cnt':>4} {'WGs':>4} {'gap':>6} {'skew':>5} | {'karg':>5} {'in med/max':>11} {'loop med/max':>12} {'st med/max':>11} {'ack med/max':>11} | {'total':>6} {'sum us':>7}"
      f" | waves: {'entry max':>9} {'loop min/med/max':>17} {'part max':>8} {'barrier':>7} {'tail':>5}")
tot_all = 0.0
for nme, rs in sorted(agg.items(), key=lambda kv: -sum(r["tot"] for r in kv[1])):
    f = lambda k: float(np.nanmean([r[k] for r in rs]))
    ssum = sum(r["tot"] for r in rs); tot_all += ssum
    print(f"{nme:44} {len(rs):4d} {rs[0]['wgs']:4d} {f('gap'):6.2f} {f('skew'):5.2f} | {f('s7m'):5.2f} {f('s1m'):5.2f}/{f('s1x'):5.2f} {f('s2m'):6.2f}/{f('s2x'):5.2f} "
          f"{f('s3m'):5.2f}/{f('s3x'):5.2f} {f('s4m'):5.2f}/{f('s4x'):5.2f} | {f('tot'):6.2f} {ssum:7.1f}", end="")
    rd = [r for r in rs if "w2" in r]
    if rd:
        g = lambda fn: float(np.nanmean([fn(r) for r in rd]))
        print(f" |        {g(lambda r: r['w0x']):9.2f} {g(lambda r: r['w2'][0]):5.2f}/{g(lambda r: r['w2'][1]):5.2f}/{g(lambda r: r['w2'][2]):5.2f} "
              f"{g(lambda r: r['w6x']):8.2f} {g(lambda r: r['w5']):7.2f} {g(lambda r: r['tail']):5.2f}")
    else:
        print()
# shader clock: attn_cp nodes carry clock64() at entry (slot 5) and exit (slot 6) beside the 100 MHz stamps (slots 0 and 4)
mhz = []
for i in range(N):
    if meta[i][0] == 1:
        v = st[i].astype(np.int64); ok = (v[:, 0] != 0) & (v[:, 5] != 0) & (v[:, 6] > v[:, 5]) & (v[:, 4] > v[:, 0])
        if ok.any():
            mhz.extend(((v[ok, 6] - v[ok, 5]) / ((v[ok, 4] - v[ok, 0]) * 0.01)).tolist())
if mhz:
    print(f"\nshader clock during the frame (clock64 ticks per us over {len(mhz)} attn_cp workgroups): median {np.median(mhz):.0f} MHz, min {np.min(mhz):.0f}, max {np.max(mhz):.0f}")
gaps = [r["gap"] for r in rows if r and np.isfinite(r["gap"])]
print(f"\nsum of node totals {tot_all:.1f} us, sum of gaps {np.nansum(gaps):.1f} us (incl. un-instrumented nodes), nodes {len(gaps) + 1}")
