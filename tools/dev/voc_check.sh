mkdir -p gpurun_out/x2
python -m pytest tests/test_bench_config_parity.py tests/test_gpu_parity.py -q -x -k "vocoder or decoder or seamless or residual or streaming" 2>&1 | tail -4 > gpurun_out/x2/tests.log
for P in 3 2; do
  Q3_CODEC_PLANES=$P bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/x2/vocoder_kernels_p$P.txt
done
cat gpurun_out/x2/tests.log; head -2 gpurun_out/x2/vocoder_kernels_p*.txt
