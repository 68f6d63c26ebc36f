# vocoder per-kernel table under extra environment switches: voc_env_ab.sh <tag> VAR=VAL...
mkdir -p gpurun_out/envab
tag=$1; shift
env "$@" bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/envab/$tag.txt
head -14 gpurun_out/envab/$tag.txt | cut -c1-110
