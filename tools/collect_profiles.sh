set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --latency-queries 50 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --latency-queries 5 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --latency-queries 5 > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bq -- python3 tools/run_batched.py --rows 12500000 --batch 1024 --path 2 --reps 10 > $O/trace_bq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_bq_a -- tools/bqbench 12500000 4 > $O/pmc_bq_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_bq_b -- tools/bqbench 12500000 4 > $O/pmc_bq_b.log 2>&1
for cfg in "1000000 10" "10000000 10" "10000000 1000" "100000000 100"; do set -- $cfg; python bench.py --rows $1 --topn $2 --steps 200 --warmup 20 --no-cpu-baseline --no-batched --latency-queries 200 >> $O/other_configs.jsonl 2>> $O/other.err; done
echo done
