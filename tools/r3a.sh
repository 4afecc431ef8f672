set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_node.py tests/test_gpu_replica.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python bench.py --virtual-shards 8 --steps 200 --warmup 20 --latency-queries 200 > $O/virtual8.json 2> $O/virtual8.err || { tail -20 $O/virtual8.err; exit 1; }
cat $O/virtual8.json
timeout -k 10 300 tools/latency 10000000 100 1000 > $O/latency.json 2> $O/latency.err || { tail -20 $O/latency.err; exit 1; }
tail -1 $O/latency.json
