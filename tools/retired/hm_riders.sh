# On the GPU box: streams of 12- and 32-query batches under (sample size, seed riders) pairs — an MI355REC_EXPERIMENTS
# build under gpurun_out/ (MI355REC_EXP_SAMPLE_LOG2, MI355REC_EXP_RIDERS); the product library is not touched.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/hmr
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_EXPERIMENTS -o $O/lib_exp.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
: > $O/riders.jsonl
for cfg in "0 8" "0 12" "0 16" "0 24" "1 16" "1 24" "1 32" "2 32" "2 48" "2 64"; do set -- $cfg
  MI355REC_EXP_SAMPLE_LOG2=$1 MI355REC_EXP_RIDERS=$2 timeout -k 10 200 python3 tools/run_half_multi.py --fp16 --lib $O/lib_exp.so --sizes "" --streams 2,12,20,32 2>> $O/err.log \
    | sed "s/^{/{\"log2\": $1, \"riders\": $2, /" >> $O/riders.jsonl
done
rm -f $O/lib_exp.so $O/cpu_backend.o
python3 - <<'PY'
import json
for l in open("gpurun_out/hmr/riders.jsonl"):
    d = json.loads(l)
    print(d["log2"], d["riders"], [(c["queries"], c["us_per_call"], c["launch_kernel_us"]) for c in d["streams"]])
PY
