# two-plane vocoder: conv geometry switches (Q3_CONV_TM4) and side-by-side utterance count
mkdir -p gpurun_out/geo
for G in 2 3 1; do
  Q3_CONV_TM4=$G Q3_CODEC_PLANES=2 bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/geo/p2_tm4_$G.txt
done
head -2 gpurun_out/geo/*.txt
