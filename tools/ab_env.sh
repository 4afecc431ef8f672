cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6env; mkdir -p $O
A="--steps 20 --warmup 5 --no-cpu-baseline --no-config0 --no-clustered --no-batched --no-single-lane --latency-queries 300"
for i in 1 2; do
python3 bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default      ', d['value'], d['value_runs'], d['p50_ms'], d['fp32_rows']['value'])"
HSA_ENABLE_INTERRUPT=0 python3 bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('no interrupt ', d['value'], d['value_runs'], d['p50_ms'], d['fp32_rows']['value'])"
done
