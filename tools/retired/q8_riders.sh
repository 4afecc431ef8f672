# On the GPU box: the streamed single-query scan over the 8-bit replica under different numbers of seed riders
# (MI355REC_EXP_RIDERS, an MI355REC_EXPERIMENTS build under gpurun_out/; the product library is not touched).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/q8r
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_EXPERIMENTS -o $O/lib_exp.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
: > $O/riders.jsonl
for R in 10000000 1000000; do
  for n in default 6 8 10 12 16 22 32; do
    if [ $n = default ]; then E="X=1"; else E="MI355REC_EXP_RIDERS=$n"; fi
    env $E timeout -k 10 200 python3 tools/run_replica.py --rows $R --topn 100 --steps 400 --check 4 --only 2 --lib $O/lib_exp.so 2>> $O/err.log \
      | sed "s/^{/{\"riders\": \"$n\", /" >> $O/riders.jsonl
  done
done
rm -f $O/lib_exp.so $O/cpu_backend.o
python3 - <<'PY'
import json
for l in open("gpurun_out/q8r/riders.jsonl"):
    d = json.loads(l)
    r = d["replica_q8"]
    print(d["riders"], d["rows"], r["us_per_step"], r["scan_kernel_us"], r["rescored_per_query"])
PY
