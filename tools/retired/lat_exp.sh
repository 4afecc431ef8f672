# On the GPU box: lone-query latency from C++ at 1 M rows x top-10 (BASELINE configs[1]) and 10 M x top-100 under the
# variants of VERDICT r3 item 8 — a bounded experiment: the default route (sample + 8-bit scan + merge), the fp32 scan
# (no sample launch), and either with fewer scanning workgroups (fewer lists for the merge).  MI355REC_EXPERIMENTS build
# under gpurun_out/; the product library is not touched.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lx
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_EXPERIMENTS -o $O/libmi355rec.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
g++ -O2 -std=c++17 -Iinclude tools/latency.cpp $P/csrc/Recommender.cpp $P/csrc/DataManager.cpp -L$O -lmi355rec -Wl,-rpath,$PWD/$O -o $O/latency
: > $O/latency.jsonl
run() {  # label rows topn replica_mode env...
  local label=$1 rows=$2 topn=$3 mode=$4; shift 4
  env "$@" timeout -k 10 120 $O/latency $rows $topn 1500 0 $mode 2>> $O/latency.err | sed "s/^{/{\"variant\": \"$label\", /" >> $O/latency.jsonl
}
for cfg in "1000000 10" "3000000 10" "10000000 100"; do set -- $cfg
  run default $1 $2 -1 X=1
  run fp32_scan $1 $2 1 X=1
  run q8_grid128 $1 $2 -1 MI355REC_EXP_REPLICA_GRID=128
  run q8_grid192 $1 $2 -1 MI355REC_EXP_REPLICA_GRID=192
  run q8_grid256 $1 $2 -1 MI355REC_EXP_REPLICA_GRID=256
  run fp32_grid256 $1 $2 1 MI355REC_EXP_FP32_GRID=256
done
python3 - <<'PY'
import json
for l in open("gpurun_out/lx/latency.jsonl"):
    d = json.loads(l)
    print(d["rows"], d["topn"], d["variant"], d["c_abi_query_row_topn"]["p50_us"], d["c_abi_query_row_topn"]["p99_us"], d["recommender_recommend_by_index"]["p50_us"])
PY
