# On the GPU box: where the multi-query pass spends what it spends per query.  Builds under gpurun_out/ with
# -DMI355_HM_EXP=1 (hits seen, not extracted), 2 (extracted, never scored), 3 (no lists written), 4 (one live column in the B operand) give WRONG results
# and exist for their timings only; the product library is not touched.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/hmx
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
for x in ${EXPS:-0 1 2 3 4}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
    -DMI355_HM_EXP=$x -o $O/lib_exp$x.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
  timeout -k 10 200 python3 tools/run_half_multi.py --fp16 --lib $O/lib_exp$x.so --sizes 1,12,32 > $O/exp$x.json 2>> $O/exp.err
done
rm -f $O/lib_exp*.so $O/cpu_backend.o
python3 - <<'PY'
import json
import os
for x in [x for x in range(5) if os.path.exists(f"gpurun_out/hmx/exp{x}.json")]:
    d = json.load(open(f"gpurun_out/hmx/exp{x}.json"))
    print("exp", x, [(c["queries"], c["pass_kernel_us"], c["rows_to_exact_chain_per_query"]) for c in d["single_calls"]],
          "streams", [(c["queries"], c["us_per_call"], c["launch_kernel_us"]) for c in d["streams"]])
PY
