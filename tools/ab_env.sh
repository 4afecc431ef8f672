# A/B on one box: the runtime's default (interrupt-driven completion signals) against HSA_ENABLE_INTERRUPT=0 (polled), the whole
# 20-step line twice each way: the headline, the batched legs over lanes, the sorted-catalogue batch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--steps 20 --warmup 5 --no-cpu-baseline --no-config0 --latency-queries 300"
for i in 1 2; do
for E in 1 0; do
HSA_ENABLE_INTERRUPT=$E python3 bench.py $A 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); b=d['batched']; c=d['clustered']['shapes']
print('HSA_ENABLE_INTERRUPT=$E', d['value'], d['value_runs'], 'p50', d['p50_ms'], 'batched', b['ms_per_call'], b['two_lanes']['ms_per_call'], b['configs4_shard']['ms_per_call'], b['configs4_shard']['two_lanes']['ms_per_call'], 'clustered batch', [s['batch_of_1024']['ms_per_call'] for s in c], 'micro', d['microbatch']['ms_per_call'], d['microbatch']['thirty_two_queries']['ms_per_call'])"
done
done
