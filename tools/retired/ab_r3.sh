# On the GPU box: the same bench lines from this tree and from a checkout of the previous round under tmp_r3/ (not
# committed; `git archive <rev> | tar -x -C tmp_r3` + build() there), alternating, to tell a regression from box variance.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab
mkdir -p $O
: > $O/ab.jsonl
for rep in 1 2; do
  for tree in . tmp_r3; do
    for cfg in "1000000 10" "10000000 100"; do set -- $cfg
      (cd $tree && python bench.py --rows $1 --topn $2 --steps 300 --warmup 30 --no-cpu-baseline --no-batched --latency-queries 100 2>> $O/err.log) \
        | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'tree':'$tree','rows':$1,'value':d['value'],'ms_per_step':d['ms_per_step'],'p50_ms':d['p50_ms'],'kernel_ms':d['roofline'].get('avg_kernel_ms'),'microbatch':d.get('microbatch',{}).get('ms_per_call')}))" >> $O/ab.jsonl
    done
  done
done
cat $O/ab.jsonl
