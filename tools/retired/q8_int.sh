# On the GPU box (round 6): the integer-dot scan over the 8-bit replica (csrc/replica_q8.hip.h).  Parity tests of the routes
# that read the replica, then the streamed / lone timings of the product build (16-bit query: six v_dot4 per row) and of the
# A/B build with an 8-bit query (-DMI355_Q8_QUERY_BITS=8: three v_dot4 per row, twice the margin), uniform and sorted rows.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/q8int
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_replica.py tests/test_gpu_parity.py tests/test_gpu_clustered.py tests/test_gpu_fuzz.py tests/test_gpu_routes.py tests/test_gpu_lanes.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355_Q8_QUERY_BITS=8 -o $O/lib_q8q8.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
: > $O/q8.jsonl
for C in "" "--catalogue clustered --contiguous --clusters 3000 --ramp"; do
  timeout -k 10 200 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 400 --check 16 $C 2>> $O/err.log | sed "s/^{/{\"query_bits\": 16, \"cat\": \"$C\", /" >> $O/q8.jsonl
  timeout -k 10 200 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 400 --check 16 --only 2 --lib $O/lib_q8q8.so $C 2>> $O/err.log | sed "s/^{/{\"query_bits\": 8, \"cat\": \"$C\", /" >> $O/q8.jsonl
done
timeout -k 10 200 python3 tools/run_replica.py --rows 1000000 --topn 10 --steps 400 --check 16 2>> $O/err.log | sed "s/^{/{\"query_bits\": 16, /" >> $O/q8.jsonl
rm -f $O/lib_q8q8.so $O/cpu_backend.o
cat $O/q8.jsonl
