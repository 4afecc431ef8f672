# Round 6: every GPU test, smoke(), then the bench line at the driver's form (20 steps) and at 300 steps.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6suite
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err || { tail -20 $O/bench20.err; exit 1; }
python bench.py --steps 300 --warmup 30 --no-config0 --no-clustered > $O/bench300.json 2> $O/bench300.err || { tail -20 $O/bench300.err; exit 1; }
python - <<'PY'
import json
for f in ("bench20", "bench300"):
    d = json.load(open(f"gpurun_out/r6suite/{f}.json"))
    print(f, json.dumps({k: d.get(k) for k in ("value", "value_runs", "ms_per_step", "p50_ms", "verified_against_oracle", "verified_queries")}))
    print(json.dumps(d["roofline"]))
    print(json.dumps(d.get("single_lane")))
    m = d["microbatch"]
    print(m["ms_per_call"], m.get("single_lane"), m.get("thirty_two_queries"))
    b = d.get("batched", {})
    print(b.get("ms_per_call"), (b.get("configs4_shard") or {}).get("ms_per_call"), (b.get("two_lanes") or {}).get("ms_per_call"))
PY
