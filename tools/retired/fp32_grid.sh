# On the GPU box: the streamed fp32 scan with fewer scanning workgroups than are resident (MI355REC_EXP_SGRID, an
# MI355REC_EXPERIMENTS build under gpurun_out/fg): three, two and a half, two, one and a half workgroups per CU.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fg
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_EXPERIMENTS -o $O/libmi355rec.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
: > $O/grid.jsonl
for rows in ${ROWS:-10000000 30000000}; do
  for g in 767 703 639 575 511 447 383; do
    MI355REC_EXP_SGRID=$g timeout -k 10 200 python3 tools/run_replica.py --rows $rows --topn 100 --only 1 --lib $O/libmi355rec.so \
      | sed "s/^{/{\"sgrid\": $g, /" >> $O/grid.jsonl
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/fg/grid.jsonl"):
    d = json.loads(l)
    print(d["sgrid"], d["rows"], d["fp32_rows"]["us_per_step"], d["fp32_rows"]["scan_kernel_us"])
PY
