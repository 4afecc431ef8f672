set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 300 tools/latency 10000000 100 1000 > $O/latency.json 2> $O/latency.err || { tail -20 $O/latency.err; exit 1; }
tail -1 $O/latency.json
timeout -k 10 300 tools/latency 1000000 10 1000 > $O/latency_1m.json 2> $O/latency.err || { tail -20 $O/latency.err; exit 1; }
tail -1 $O/latency_1m.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/debug_hm.py > $O/trace.log 2>&1 || { tail -30 $O/trace.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/r3e/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if "mi355::" in r["Name"]:
        print(r["Name"].split("(")[0][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
