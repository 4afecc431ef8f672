set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5y; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_merge.py tests/test_gpu_parity.py tests/test_gpu_clustered.py tests/test_gpu_replica.py -x -q -m gpu > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for shape in uniform-bound full cluster-65 sparse-8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/new_$shape -- python3 tools/merge_ab.py --shape $shape > $O/new_$shape.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
for d in sorted(glob.glob(o + "/*_*/")):
    for f in glob.glob(d + "**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "merge_kernel" in r["Name"]:
                print(d.split("/")[-2], r["Calls"], round(float(r["AverageNs"]) / 1000, 2), "us")
PY
python3 tools/run_half_multi.py --fp16 --sizes 12 --streams 12 --calls 40 2>/dev/null | cut -c1-600
