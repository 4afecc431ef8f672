# On the GPU box: merge_kernel's own duration (rocprofv3 kernel trace) per list shape, this tree's library against another
# (tools/_ab/*.so, built from an earlier commit on the build container).   bash tools/merge_ab.sh tools/_ab/libr4.so out-dir
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OTHER=$1; O=${2:-gpurun_out/merge_ab}; mkdir -p $O
for shape in uniform-bound full top10 cluster-65 sparse-8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/new_$shape -- python3 tools/merge_ab.py --shape $shape > $O/new_$shape.log 2>&1
  MI355REC_CAPI_LENIENT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/old_$shape -- python3 tools/merge_ab.py --shape $shape --lib $OTHER > $O/old_$shape.log 2>&1
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
