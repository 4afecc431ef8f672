# Kernel time of the single-query scans against catalogue size: fixed cost vs per-tile cost.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q8
: > gpurun_out/q8/sweep.jsonl
for R in 1000000 2000000 4000000 10000000; do
  timeout -k 10 200 python tools/run_replica.py --rows $R --topn 10 --steps 400 --check 8 >> gpurun_out/q8/sweep.jsonl 2>gpurun_out/q8/sweep.err
done
timeout -k 10 200 python tools/run_replica.py --rows 10000000 --topn 100 --steps 400 --check 8 >> gpurun_out/q8/sweep.jsonl 2>>gpurun_out/q8/sweep.err
cat gpurun_out/q8/sweep.jsonl
