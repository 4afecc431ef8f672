# Lone-query latency from C++ (tools/latency.cpp) at 10 M / top-100, 1 M / top-10, 3 M / top-10.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lat
g++ -O2 -std=c++17 -Iinclude tools/latency.cpp spotify_recommender_amd/csrc/Recommender.cpp spotify_recommender_amd/csrc/DataManager.cpp \
  -Lspotify_recommender_amd -lmi355rec -Wl,-rpath,'$ORIGIN/../spotify_recommender_amd' -o tools/latency
: > gpurun_out/lat/latency.jsonl
for cfg in "10000000 100" "1000000 10" "3000000 10"; do set -- $cfg
  timeout -k 10 120 tools/latency $1 $2 2000 >> gpurun_out/lat/latency.jsonl 2>> gpurun_out/lat/latency.err
done
cat gpurun_out/lat/latency.jsonl
