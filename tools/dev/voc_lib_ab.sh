# vocoder per-kernel table for an alternative build: voc_lib_ab.sh <lib.so> <planes> <tag>
mkdir -p gpurun_out/libab
Q3TTS_LIB=$PWD/$1 Q3_CODEC_PLANES=$2 bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/libab/$3.txt
head -16 gpurun_out/libab/$3.txt | cut -c1-110
