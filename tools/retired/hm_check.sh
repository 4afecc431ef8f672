# The multi-query pass after a change: its tests, then its timings over both replicas; SWEEP=1 adds the sample-size
# sweep (an MI355REC_EXPERIMENTS build under gpurun_out/: regions of 1024 << l rows for l = 0, 1, 2).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/hm
mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_half_multi.py tests/test_gpu_fuzz.py tests/test_gpu_routes.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python3 tools/run_half_multi.py --fp16 > $O/fp16.json 2> $O/fp16.err || { tail -20 $O/fp16.err; exit 1; }
timeout -k 10 200 python3 tools/run_half_multi.py > $O/q8.json 2> $O/q8.err || { tail -20 $O/q8.err; exit 1; }
if [ -n "$SWEEP" ]; then
  P=spotify_recommender_amd
  g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
    -DMI355REC_EXPERIMENTS -o $O/libmi355rec_exp.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
  for l in 0 1 2; do
    MI355REC_EXP_SAMPLE_LOG2=$l timeout -k 10 200 python3 tools/run_half_multi.py --fp16 --lib $O/libmi355rec_exp.so --sizes 4,8,12,16,32 > $O/sweep_l$l.json 2>> $O/sweep.err
  done
fi
python3 - <<'PY'
import json, os
names = ["fp16", "q8"] + [f"sweep_l{l}" for l in (0, 1, 2) if os.path.exists(f"gpurun_out/hm/sweep_l{l}.json")]
for f in names:
    d = json.load(open(f"gpurun_out/hm/{f}.json"))
    print(f, [(c["queries"], c["us_per_call"], c["pass_kernel_us"], c["rows_to_exact_chain_per_query"]) for c in d["single_calls"]])
    print(f, "streams", [(c["queries"], c["us_per_call"], c["launch_kernel_us"]) for c in d["streams"]])
PY
