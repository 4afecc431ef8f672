set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for W in 16 32; do
timeout -k 10 300 python bench.py --virtual-shards 8 --window $W --steps 320 --warmup 32 --latency-queries 100 > $O/virtual8_w$W.json 2> $O/virtual8.err || { tail -20 $O/virtual8.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r3d/virtual8_w$W.json"))
print("window $W:", json.dumps({k: d[k] for k in ("value", "ms_per_step", "p50_ms", "host", "single_engine_same_gpu", "verified_against_oracle")}))
print("  roofline", json.dumps(d["roofline"]))
PY
done
