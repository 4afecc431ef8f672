# On the GPU box: where the routes' thresholds belong.  Lone synchronous queries (tools/latency.cpp) with the replica
# forced ON (2) and OFF (1) at sizes around kLoneFp32MaxRows, and streamed queries (bench.py) (tools/run_replica.py) over the fp32 rows and
# over the replicas at sizes around kHalfAutoMinRows.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/thr
mkdir -p $O
g++ -O2 -std=c++17 -Iinclude tools/latency.cpp spotify_recommender_amd/csrc/Recommender.cpp spotify_recommender_amd/csrc/DataManager.cpp \
  -Lspotify_recommender_amd -lmi355rec -Wl,-rpath,$PWD/spotify_recommender_amd -o $O/latency
: > $O/lone.jsonl
for R in 200000 400000 700000 1000000 1400000 2000000 3000000; do
  for M in 1 2; do
    $O/latency $R 10 1500 0 $M 2>> $O/err.log | grep '^{' >> $O/lone.jsonl
  done
done
rm -f $O/latency
: > $O/stream.jsonl
for R in 100000 200000 400000 700000 1000000; do
  python3 tools/run_replica.py --rows $R --topn 10 --steps 400 --check 4 >> $O/stream.jsonl 2>> $O/err.log
done
python3 - <<'PY'
import json
for l in open("gpurun_out/thr/lone.jsonl"):
    d = json.loads(l)
    print("lone", d["rows"], "mode", d["replica_mode"], d["c_abi_query_row_topn"]["p50_us"])
for l in open("gpurun_out/thr/stream.jsonl"):
    d = json.loads(l)
    print("stream", d["rows"], "fp32", d["fp32_rows"]["us_per_step"], "q8", d["replica_q8"]["us_per_step"], "fp16", d["replica_fp16"]["us_per_step"],
          "| lone p50: fp32", d["fp32_rows"]["p50_us"], "q8", d["replica_q8"]["p50_us"])
PY
