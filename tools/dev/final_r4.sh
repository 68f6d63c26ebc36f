# end-of-round evidence: vocoder MFMA counters in both modes, whole GPU suite, the default bench line
mkdir -p gpurun_out/final
bash tools/pmc_vocoder.sh 640 > /dev/null 2>&1; cp gpurun_out/pmc/vocoder_mfma_T640.txt gpurun_out/final/pmc_vocoder_mfma_T640.txt
Q3_CODEC_PLANES=2 bash tools/pmc_vocoder.sh 640 > /dev/null 2>&1; cp gpurun_out/pmc/vocoder_mfma_T640.txt gpurun_out/final/pmc_vocoder_mfma_T640_2planes.txt
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/final/suite.txt
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -3 gpurun_out/final/suite.txt
