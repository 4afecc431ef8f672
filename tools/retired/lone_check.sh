# On the GPU box: the tests that exercise lone synchronous queries, then C++ latencies (tools/latency.cpp) at the sizes
# of BASELINE configs[1] and around the routes' thresholds.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lone
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_routes.py tests/test_gpu_shim.py tests/test_gpu_replica.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
g++ -O2 -std=c++17 -Iinclude tools/latency.cpp spotify_recommender_amd/csrc/Recommender.cpp spotify_recommender_amd/csrc/DataManager.cpp \
  -Lspotify_recommender_amd -lmi355rec -Wl,-rpath,$PWD/spotify_recommender_amd -o $O/latency
: > $O/latency.jsonl
for cfg in "114000 10" "500000 10" "1000000 10" "1000000 100" "1400000 10" "1600000 10" "10000000 100"; do set -- $cfg
  $O/latency $1 $2 2000 2>> $O/latency.err | grep '^{' >> $O/latency.jsonl
done
rm -f $O/latency
python3 - <<'PY'
import json
for l in open("gpurun_out/lone/latency.jsonl"):
    d = json.loads(l)
    print(d["rows"], d["topn"], d["c_abi_query_row_topn"], d["recommender_recommend_by_index"]["p50_us"], d["recommender_matches_c_abi"])
PY
