#!/bin/bash
# Collects HBM traffic PMC counters (FETCH_SIZE, WRITE_SIZE in separate passes, as MI355X_MICROARCH.md §HBM
# prescribes) for every GEMV shape of one Qwen3-TTS-1.7B frame at M = $1 (default 8). Run ON the GPU box:
#   bash tools/pmc_collect.sh 8   → gpurun_out/pmc/pmc_gemv_M8.json
M=${1:-8}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# name N K epilogue fused-norm tiling (-1 = the engine's unsplit choice, 3 = split-K in two: what the frame loop launches for the o / down projections at 3 <= M <= 16)
SK=-1; if [ "$M" -ge 3 ] && [ "$M" -le 16 ]; then SK=3; fi
SHAPES=("talker_qkv 4096 2048 0 1 -1" "talker_o 2048 2048 1 0 $SK" "talker_gateup 6144 2048 3 1 -1" "talker_down 2048 6144 1 0 $SK" "codec_head 3072 2048 0 0 -1" "cp_qkv 4096 1024 0 1 -1" "cp_o 1024 2048 1 0 $SK" "cp_gateup 3072 1024 3 1 -1" "cp_down 1024 3072 1 0 $SK" "cp_lm_head 2048 1024 0 1 -1" "cp_mtp_proj 1024 2048 0 0 -1")
for shape in "${SHAPES[@]}"; do set -- $shape
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/$1_$ctr" -o p -- python "$ROOT/tools/pmc_gemv.py" $@ $M > "$OUT/$1_$ctr.log" 2>&1
  done
done
cd "$OUT" && python - "$M" <<'PY'
import csv, glob, json, sys, collections
M = int(sys.argv[1]); out = {}
for f in sorted(glob.glob("*/p_counter_collection.csv")):
    shape, ctr = f.split("/")[0].rsplit("_", 2)[0], "_".join(f.split("/")[0].rsplit("_", 2)[1:])
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemv" in r["Kernel_Name"]]
    if vals: out.setdefault(shape, {})[ctr] = sum(vals) / len(vals)
dims = {"talker_qkv": (4096, 2048, 0), "talker_o": (2048, 2048, 1), "talker_gateup": (6144, 2048, 3), "talker_down": (2048, 6144, 1), "codec_head": (3072, 2048, 0),
        "cp_qkv": (4096, 1024, 0), "cp_o": (1024, 2048, 1), "cp_gateup": (3072, 1024, 3), "cp_down": (1024, 3072, 1), "cp_lm_head": (2048, 1024, 0), "cp_mtp_proj": (1024, 2048, 0)}
res = {}
for shape, d in out.items():
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams → x2
    res[shape] = {"fetch_bytes_corrected": d.get("FETCH_SIZE", 0) * 1024 * 2, "write_bytes": d.get("WRITE_SIZE", 0) * 1024,
                  "fetch_size_raw_kib": d.get("FETCH_SIZE"), "write_size_raw_kib": d.get("WRITE_SIZE"),
                  "N": dims[shape][0], "K": dims[shape][1], "epi": dims[shape][2], "algorithmic_bytes": dims[shape][0] * dims[shape][1] * 2 * (2 if dims[shape][2] == 3 else 1)}
json.dump({"M": M, "note": "per-launch means over 304 launches; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction (MI355X_MICROARCH.md §HBM)", "shapes": res}, open(f"pmc_gemv_M{M}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*kernel_trace.csv" -delete
