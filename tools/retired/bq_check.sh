# On the GPU box: the batched path's tests, then its timings and per-kernel times at 10 M and 12.5 M rows x 1024 queries.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bqc
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_routes.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for R in 10000000 12500000; do
  python3 tools/run_batched.py --rows $R --batch 1024 --reps 30 --path 2 2>> $O/err.log | grep '^{' | tee $O/run_$R.json | cut -c1-260
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$R -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 20 --path 2 > $O/trace_$R.log 2>&1
  python3 - <<PY
import csv, glob, os
f = max(glob.glob("$O/trace_$R/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "bq_" in n or "queued" in n:
        print(f'{float(r["AverageNs"])/1e3:9.1f} us x {r["Calls"]:>4}  {n[:90]}')
PY
done
