# Round-5 evidence runs (one gpurun call per PART): the bench line, rocprofv3 kernel stats of the same command, PMC
# passes (one counter set per run, never together with a trace).  Outputs under gpurun_out/r5p/; summarised into profiles/r05_*
# by tools/summarize_profiles.py (kernel stats + HBM traffic), tools/summarize_pmc.py (SQ counters) and tools/summarize_r05.py
# (the multi-query pass's traffic split, the clustered catalogues).
#   PART=A  the default bench line, its kernel trace, its FETCH_SIZE / WRITE_SIZE passes
#   PART=B  the multi-query pass: kernel trace + FETCH / WRITE of a STREAM of 12-query batches and of single calls (the pass on
#           its own, the sample launch, the merge launch: what a streamed launch carries for its neighbours, piece by piece)
#   PART=C  the batched two-pass path at 12.5 M rows: kernel trace, SQ counters, FETCH / WRITE
#   PART=D  contiguous clusters on every route (tools/clustered.sh), C++ latency, the other configs, virtual shards
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p
mkdir -p $O
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
case ${PART:-A} in
A)
  python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-config0 --no-clustered --no-single-lane --latency-queries 50 > $O/trace.log 2>&1
  echo "trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-config0 --no-clustered --no-single-lane --no-c5-shard --latency-queries 5 > $O/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-config0 --no-clustered --no-single-lane --no-c5-shard --latency-queries 5 > $O/write.log 2>&1
  ;;
B)
  S="python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60"
  C="python3 tools/run_half_multi.py --fp16 --sizes 12 --streams= --calls 60"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hm_trace_stream -- $S > $O/hm_trace_stream.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hm_trace_call -- $C > $O/hm_trace_call.log 2>&1
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CTR --output-format csv -d $O/hm_${CTR}_stream -- $S > $O/hm_${CTR}_stream.log 2>&1
    rocprofv3 --pmc $CTR --output-format csv -d $O/hm_${CTR}_call -- $C > $O/hm_${CTR}_call.log 2>&1
  done
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/hm_sq_stream -- $S > $O/hm_sq_stream.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hm_trace_stream32 -- python3 tools/run_half_multi.py --fp16 --only-stream 32 --calls 60 > $O/hm_trace_stream32.log 2>&1
  python3 tools/run_half_multi.py --fp16 > $O/half_multi.json 2> $O/half_multi.err
  ;;
C)
  R=12500000
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bq_trace -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 20 > $O/bq_trace.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bq_trace10 -- python3 tools/run_batched.py --rows 10000000 --batch 1024 --reps 20 > $O/bq_trace10.log 2>&1
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/bq_sq_a -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 > $O/bq_sq_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/bq_sq_b -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 > $O/bq_sq_b.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bq_fetch -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 > $O/bq_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bq_write -- python3 tools/run_batched.py --rows $R --batch 1024 --reps 6 > $O/bq_write.log 2>&1
  ;;
D)
  bash tools/clustered.sh $O/cl > $O/cl.log 2>&1
  echo "clustered done"
  g++ -O2 -std=c++17 -Iinclude tools/latency.cpp spotify_recommender_amd/csrc/Recommender.cpp spotify_recommender_amd/csrc/DataManager.cpp \
    -Lspotify_recommender_amd -lmi355rec -Wl,-rpath,$PWD/spotify_recommender_amd -o $O/latency
  $O/latency 10000000 100 2000 2> $O/latency.err | grep '^{' > $O/latency_10m.json
  $O/latency 1000000 10 2000 2>> $O/latency.err | grep '^{' > $O/latency_1m.json
  rm -f $O/latency
  for cfg in "1000000 10" "10000000 10" "10000000 1000" "100000000 100"; do set -- $cfg; python bench.py --rows $1 --topn $2 --steps 200 --warmup 20 --no-cpu-baseline --no-batched --no-config0 --no-clustered --latency-queries 200 >> $O/other_configs.jsonl 2>> $O/other.err; done
  python bench.py --virtual-shards 8 --no-cpu-baseline --no-c5-shard > $O/virtual8.json 2> $O/virtual8.err
  python bench.py --virtual-shards 8 --placement replicated --no-cpu-baseline --no-c5-shard > $O/virtual8_replicated.json 2>> $O/virtual8.err
  ;;
esac
echo done
