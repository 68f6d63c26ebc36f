# vocoder with three / two bf16 planes per operand: parity tests, per-kernel table of one 640-frame decode, bench lines
mkdir -p gpurun_out/x2
python -m pytest tests/test_bench_config_parity.py -q -x -k "vocoder" 2>&1 | tail -5 > gpurun_out/x2/tests.log
for P in 3 2; do
  Q3_CODEC_PLANES=$P bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/x2/vocoder_kernels_p$P.txt
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/x2/bench.json 2> gpurun_out/x2/bench.err
