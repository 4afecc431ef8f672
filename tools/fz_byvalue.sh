# The randomised parity sweep with queries by value (round 6), two big seeds; then the one-process-per-GPU bench path forced
# through its RCCL code on one GPU (BENCH_FORCE_SHARDED=1: a one-rank all-gather per window, the `group` object).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fz2
for SEED in 11 4242; do
  FUZZ_SEED=$SEED FUZZ_CASES=400 FUZZ_CASES2=20 timeout -k 10 560 python -m pytest tests/test_gpu_fuzz.py::test_randomised_parity -x -q -m gpu > gpurun_out/fz2/fuzz_$SEED.log 2>&1 || { tail -40 gpurun_out/fz2/fuzz_$SEED.log; exit 1; }
  tail -2 gpurun_out/fz2/fuzz_$SEED.log
done
BENCH_FORCE_SHARDED=1 timeout -k 10 400 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-config0 --no-clustered --no-c5-shard --latency-queries 50 > gpurun_out/fz2/forced_sharded.json 2> gpurun_out/fz2/forced_sharded.err || { tail -30 gpurun_out/fz2/forced_sharded.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/fz2/forced_sharded.json"))
print(d["value"], d["value_runs"], json.dumps(d.get("group"))[:900])
print(json.dumps(d.get("transport"))[:600])
PY
