# Round 6: queries BY VALUE (nothing excluded) on sorted catalogues against the same rows by index, every single-query route;
# the new tests; then the batched path at 10 M / 12.5 M rows.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bv
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_clustered.py tests/test_gpu_lanes.py tests/test_gpu_batched.py tests/test_gpu_half_multi.py tests/test_gpu_fuzz.py tests/test_gpu_node.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
: > $O/bv.jsonl
for C in "--clusters 3000 --spread 0.03" "--clusters 300 --spread 0.01"; do
  for V in "" "--by-value"; do
    timeout -k 10 200 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300 --check 4 --catalogue clustered --contiguous --ramp $C $V 2>> $O/err.log >> $O/bv.jsonl
  done
done
timeout -k 10 200 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300 --check 4 --by-value 2>> $O/err.log >> $O/bv.jsonl
timeout -k 10 200 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300 --check 4 2>> $O/err.log >> $O/bv.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r6bv/bv.jsonl"):
    d = json.loads(l)
    print(d.get("by_value"), {k: (v["us_per_step"], v["scan_kernel_us"], v["rescored_per_query"], v["p50_us"]) for k, v in d.items() if isinstance(v, dict)})
PY
for R in 10000000 12500000; do timeout -k 10 300 python3 tools/run_batched.py --rows $R --batch 1024 --reps 20 2>> $O/err.log | tail -1 | cut -c1-600; done
# the stream of single queries from a C++ host over 1, 2, 3 lanes (no Python in the loop)
g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/lanes.cpp -Lspotify_recommender_amd -lmi355rec -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/spotify_recommender_amd -o $O/lanes
timeout -k 10 300 $O/lanes 10000000 100 3000 3 2>> $O/err.log | tail -5
rm -f $O/lanes
