# On the GPU box: where pass 2 of the batched path spends its time (MI355REC_EXPERIMENTS build, kernel trace per variant).
#   exp 0: as shipped   exp 1: the new loop, no tile counts as visited   exp 2: every visited tile skips all its blocks
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bqx
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_EXPERIMENTS ${EXTRA_DEFS} -o $O/libmi355rec_exp.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
R=${ROWS:-10000000}
for X in ${EXPS:-0 1 2}; do
  MI355REC_BQ_EXP=$X rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_x$X -- python3 tools/run_batched.py --lib $O/libmi355rec_exp.so --rows $R --batch 1024 --reps 20 --path 2 > $O/trace_x$X.log 2>&1
  python3 - <<PY
import csv, glob, os
f = max(glob.glob("$O/trace_x$X/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
print("== exp $X rows $R")
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "bq_pass" in n:
        print(f'{float(r["AverageNs"])/1e3:9.1f} us x {r["Calls"]:>4}  {n[:80]}')
PY
done
