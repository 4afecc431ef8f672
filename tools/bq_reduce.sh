# VERDICT r5 item 3b, the one variant: pass 2's hit test on packed signs (-DMI355_BQ_REDUCE=1: 16 v_cvt_pkrtz + 8 three-input ANDs
# per two MFMAs) against the product's v_max3_i32 tree, same box: the batch's time, the pass kernels alone (rocprofv3) and the vector
# instructions per MFMA (PMC).  The variant library is built under gpurun_out/ and loaded with --lib; the product library is not touched.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bqred
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355_BQ_REDUCE=1 -o $O/lib_red.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
R="--rows 12500000 --batch 1024"
for L in product variant; do
  if [ $L = variant ]; then LIB="--lib $O/lib_red.so"; else LIB=""; fi
  for i in 1 2; do timeout -k 10 200 python3 tools/run_batched.py $R --reps 20 $LIB 2>> $O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'ms_per_batch', d['ms_per_batch'], 'pass_kernel_ms', d['pass_kernel_ms'], 'candidates', d['candidates_total'])"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$L -- python3 tools/run_batched.py $R --reps 20 $LIB > $O/trace_$L.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a_$L -- python3 tools/run_batched.py $R --reps 6 $LIB > $O/pmc_a_$L.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_b_$L -- python3 tools/run_batched.py $R --reps 6 $LIB > $O/pmc_b_$L.log 2>&1
done
rm -f $O/lib_red.so $O/cpu_backend.o
python3 - <<'PY'
import collections, csv, glob, json, os
O = "gpurun_out/bqred"
out = {}
for L in ("product", "variant"):
    e = {}
    f = max(glob.glob(f"{O}/trace_{L}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if "bq_pass_kernel<32" in r["Name"]:
            e["pass2_us" if "<32, true" in r["Name"] else "pass1_us"] = round(float(r["AverageNs"]) / 1e3, 1)
    c = collections.defaultdict(list)
    for d in ("pmc_a", "pmc_b"):
        f = max(glob.glob(f"{O}/{d}_{L}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        for r in csv.DictReader(open(f)):
            if "bq_pass_kernel<32, true" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in c.items()}
    e["pass2_valu_per_mfma"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 2)
    e["pass2_gpu_cycles_per_mfma_per_simd"] = round((m["GRBM_GUI_ACTIVE"] / 8) / (m["SQ_INSTS_MFMA"] / 1024), 1)
    out[L] = e
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
