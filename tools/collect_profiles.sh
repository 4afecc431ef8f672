# Round-3 evidence run (one gpurun call): bench lines, rocprofv3 kernel stats, PMC passes.  Outputs under
# gpurun_out/r3p/; tools/summarize_profiles.py / summarize_pmc.py turn them into the files kept under profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --latency-queries 50 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --latency-queries 5 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --latency-queries 5 > $O/write.log 2>&1
echo "bench profiles done"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_q8 -- python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 120 --check 8 --only 2 > $O/pmc_q8.log 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_half -- python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 120 --check 8 --only 3 > $O/pmc_half.log 2>&1
python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300 > $O/replica_ab.json 2> $O/replica_ab.err
python3 tools/run_half_multi.py --fp16 > $O/half_multi.json 2> $O/half_multi.err
python3 tools/run_half_multi.py > $O/half_multi_q8.json 2>> $O/half_multi.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_hm -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/trace_hm.log 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_hm_a -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_hm_b -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_hm_f -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_hm_w -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_w.log 2>&1
echo "kernel profiles done"
python bench.py --virtual-shards 8 --no-cpu-baseline > $O/virtual8.json 2> $O/virtual8.err
g++ -O2 -std=c++17 -Iinclude tools/latency.cpp spotify_recommender_amd/csrc/Recommender.cpp spotify_recommender_amd/csrc/DataManager.cpp \
  -Lspotify_recommender_amd -lmi355rec -Wl,-rpath,'$ORIGIN/../spotify_recommender_amd' -o tools/latency
tools/latency 10000000 100 2000 > $O/latency_10m.json 2> $O/latency.err
tools/latency 1000000 10 2000 > $O/latency_1m.json 2>> $O/latency.err
for cfg in "1000000 10" "10000000 10" "10000000 1000" "100000000 100"; do set -- $cfg; python bench.py --rows $1 --topn $2 --steps 200 --warmup 20 --no-cpu-baseline --no-batched --latency-queries 200 >> $O/other_configs.jsonl 2>> $O/other.err; done
echo done
