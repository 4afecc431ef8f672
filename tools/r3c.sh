set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_half_multi.py tests/test_gpu_replica.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 250 python tools/debug_hm.py > $O/timing.log 2>&1 || { tail -30 $O/timing.log; exit 1; }
cat $O/timing.log
