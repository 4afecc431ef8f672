# On the GPU box: pass 2's inner loop under the scheduling experiments of csrc/batched.hip.h (MI355_BQ_SCHED = 0 product,
# 1 s_setprio around the MFMAs, 2 sched_group_barrier interleave, 3 both) — builds under gpurun_out/, never the product
# library; 10 M and 12.5 M rows x 1024 queries x top-100 through tools/run_batched.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/bqs}
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
: > $O/sched.jsonl
for V in ${VARIANTS:-0 1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
    -DMI355_BQ_SCHED=$V -o $O/lib_s$V.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp 2> $O/build$V.log
  for R in 10000000 12500000; do
    timeout -k 10 120 python3 tools/run_batched.py --lib $O/lib_s$V.so --rows $R --batch 1024 --reps 20 --check 8 \
      | sed "s/^{/{\"sched\": $V, /" >> $O/sched.jsonl
  done
  rm -f $O/lib_s$V.so
done
rm -f $O/cpu_backend.o
python3 - $O <<'PY'
import json, sys
for line in open(sys.argv[1] + "/sched.jsonl"):
    d = json.loads(line)
    print("sched", d["sched"], "rows", d["rows"], "ms", d["ms_per_batch"], "pass_ms", d["pass_kernel_ms"], "same", d.get("matches_single_query_fp32_scan"))
PY
