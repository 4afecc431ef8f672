# Round-6 evidence runs (one gpurun call per PART).  Outputs under gpurun_out/r6p/, summarised into profiles/r06_* by
# tools/summarize_r06.py.  Every rocprofv3 command has python3 directly after `--`; PMC passes are their own runs.
#   PART=A  the default bench line; its kernel trace over the TWO lanes it runs on; the same command with --lanes 1 (ONE handle:
#           every streamed kernel alone on the chip — the durations DESIGN §5 quotes); FETCH_SIZE / WRITE_SIZE passes (one handle)
#   PART=C  the batched two-pass path at 12.5 M and 10 M rows: kernel trace, SQ counters
#   PART=D  contiguous clusters on every route, queries by value, the other configs, virtual shards
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p
mkdir -p $O
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
B="--no-cpu-baseline --no-config0 --no-clustered --no-single-lane --latency-queries 50"
case ${PART:-A} in
A)
  python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench done"
  python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py $B > $O/trace.log 2>&1
  echo "trace (two lanes) done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- python3 bench.py --lanes 1 $B > $O/trace1.log 2>&1
  echo "trace (one handle) done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --lanes 1 --steps 40 --warmup 5 $B --no-c5-shard > $O/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --lanes 1 --steps 40 --warmup 5 $B --no-c5-shard > $O/write.log 2>&1
  ;;
C)
  for R in 12500000 10000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/bq_trace_$R -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 20 > $O/bq_trace_$R.log 2>&1
  done
  R=12500000
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/bq_sq_a -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 > $O/bq_sq_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/bq_sq_b -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 > $O/bq_sq_b.log 2>&1
  ;;
D)
  python bench.py --catalogue clustered-contiguous --steps 100 --warmup 10 --no-cpu-baseline --no-config0 --no-c5-shard --latency-queries 100 > $O/clustered.json 2> $O/clustered.err
  for cfg in "1000000 10" "10000000 10" "10000000 1000" "100000000 100"; do set -- $cfg; python bench.py --rows $1 --topn $2 --steps 200 --warmup 20 --no-cpu-baseline --no-batched --no-config0 --no-clustered --latency-queries 200 >> $O/other_configs.jsonl 2>> $O/other.err; done
  python bench.py --virtual-shards 8 --no-cpu-baseline --no-c5-shard > $O/virtual8.json 2> $O/virtual8.err
  python bench.py --virtual-shards 8 --placement replicated --no-cpu-baseline --no-c5-shard > $O/virtual8_replicated.json 2>> $O/virtual8.err
  python bench.py --virtual-shards 2 --placement replicated --no-cpu-baseline --no-c5-shard > $O/virtual2_replicated.json 2>> $O/virtual8.err
  ;;
esac
echo done
