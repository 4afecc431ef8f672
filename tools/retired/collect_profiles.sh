# Round-4 evidence runs (one gpurun call per PART): bench lines, rocprofv3 kernel stats, PMC passes.  Outputs under
# gpurun_out/r4p/; tools/summarize_profiles.py / summarize_pmc.py / summarize_batched_profiles.py turn them into the
# files kept under profiles/.
#   PART=A  the default bench line, its kernel trace, its FETCH_SIZE / WRITE_SIZE passes
#   PART=B  SQ counters of the single-query scans, of the fp32 multi-query pass and of the multi-query pass over the replica
#   PART=C  the batched two-pass path: kernel traces with and without tile skipping, SQ counters, FETCH / WRITE
#   PART=D  virtual shards (sharded and replicated), C++ latency, the other configs
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p
mkdir -p $O
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"
case ${PART:-A} in
A)
  python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --latency-queries 50 > $O/trace.log 2>&1
  echo "trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-c5-shard --latency-queries 5 > $O/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-c5-shard --latency-queries 5 > $O/write.log 2>&1
  ;;
B)
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_q8 -- python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 120 --check 8 --only 2 > $O/pmc_q8.log 2>&1
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_half -- python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 120 --check 8 --only 3 > $O/pmc_half.log 2>&1
  echo "single-query scans done"
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_multi_a -- python3 tools/run_multi_pass.py > $O/pmc_multi_a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_multi_b -- python3 tools/run_multi_pass.py > $O/pmc_multi_b.log 2>&1
  echo "fp32 multi-query pass done"
  python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300 > $O/replica_ab.json 2> $O/replica_ab.err
  python3 tools/run_half_multi.py --fp16 > $O/half_multi.json 2> $O/half_multi.err
  python3 tools/run_half_multi.py > $O/half_multi_q8.json 2>> $O/half_multi.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_hm -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/trace_hm.log 2>&1
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_hm_a -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_a.log 2>&1
  rocprofv3 --pmc $SQ2 --output-format csv -d $O/pmc_hm_b -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_b.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_hm_f -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_hm_w -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_w.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_hm32 -- python3 tools/run_half_multi.py --fp16 --only-stream 32 --calls 60 > $O/trace_hm32.log 2>&1
  ;;
C)
  for R in 10000000 12500000; do
    for PV in 2 5; do
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/bq_trace_${R}_p$PV -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 20 --path $PV > $O/bq_trace_${R}_p$PV.log 2>&1
    done
  done
  echo "batched traces done"
  R=12500000
  for PV in 2 5; do
    rocprofv3 --pmc $SQ1 --output-format csv -d $O/bq_pmc_a_p$PV -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 --path $PV > $O/bq_pmc_a_p$PV.log 2>&1
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/bq_pmc_b_p$PV -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 --path $PV > $O/bq_pmc_b_p$PV.log 2>&1
  done
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bq_pmc_f -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 --path 2 > $O/bq_pmc_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bq_pmc_w -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 --path 2 > $O/bq_pmc_w.log 2>&1
  ;;
D)
  python bench.py --virtual-shards 8 --no-cpu-baseline --no-c5-shard > $O/virtual8.json 2> $O/virtual8.err
  python bench.py --virtual-shards 8 --placement replicated --no-cpu-baseline --no-c5-shard > $O/virtual8_replicated.json 2>> $O/virtual8.err
  echo "virtual shards done"
  g++ -O2 -std=c++17 -Iinclude tools/latency.cpp spotify_recommender_amd/csrc/Recommender.cpp spotify_recommender_amd/csrc/DataManager.cpp \
    -Lspotify_recommender_amd -lmi355rec -Wl,-rpath,$PWD/spotify_recommender_amd -o $O/latency
  $O/latency 10000000 100 2000 2> $O/latency.err | grep '^{' > $O/latency_10m.json
  $O/latency 1000000 10 2000 2>> $O/latency.err | grep '^{' > $O/latency_1m.json
  rm -f $O/latency
  for cfg in "1000000 10" "10000000 10" "10000000 1000" "100000000 100"; do set -- $cfg; python bench.py --rows $1 --topn $2 --steps 200 --warmup 20 --no-cpu-baseline --no-batched --latency-queries 200 >> $O/other_configs.jsonl 2>> $O/other.err; done
  ;;
esac
echo done
