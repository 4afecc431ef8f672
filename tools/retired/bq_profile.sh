# On the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) and SQ counters of the batched two-pass path,
# with (path 2) and without (path 5) tile skipping.  Outputs under gpurun_out/bqp/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bqp
mkdir -p $O
for R in ${ROWS:-10000000 12500000}; do
  for PV in 2 5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${R}_p$PV -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 20 --path $PV > $O/trace_${R}_p$PV.log 2>&1
    python3 - <<PY
import csv, glob, os
f = max(glob.glob("$O/trace_${R}_p$PV/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
print("== rows $R path $PV")
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "bq_" in n or "queued" in n:
        print(f'{float(r["AverageNs"])/1e3:9.1f} us x {r["Calls"]:>4}  {n[:110]}')
PY
  done
done
if [ -n "$PMC" ]; then
  R=${PMC_ROWS:-12500000}
  for PV in 2 5; do
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a_p$PV -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 --path $PV > $O/pmc_a_p$PV.log 2>&1
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_b_p$PV -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 --path $PV > $O/pmc_b_p$PV.log 2>&1
  done
fi
echo done
