set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_half_multi.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests2.log 2>&1 || { tail -60 $O/tests2.log; exit 1; }
tail -3 $O/tests2.log
timeout -k 10 400 python bench.py --steps 100 --warmup 10 --latency-queries 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3b/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "p50_ms", "microbatch")}, indent=1))
PY
