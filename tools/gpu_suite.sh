# The round-end checks in one gpurun call: every GPU test, smoke(), a short bench line.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/suite
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 --latency-queries 100 > $O/bench20.json 2> $O/bench20.err || { tail -20 $O/bench20.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/suite/bench20.json"))
print(json.dumps({k: d.get(k) for k in ("value", "ms_per_step", "p50_ms", "verified_against_oracle", "verified_queries")}), d["roofline"]["frac"], d["roofline"].get("survey_frac"), d["microbatch"]["ms_per_call"])
PY
