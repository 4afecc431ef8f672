# On the GPU box: the batched two-pass path and the multi-query pass over catalogues that are not uniform noise
# (tools/catalogues.py): time per call, what goes to the exact chain / the exact queue, keys against the fp32 single-query scan.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cl
mkdir -p $O
: > $O/batched.jsonl
: > $O/multi.jsonl
for cfg in "3000 0.1" "3000 0.03" "3000 0.01" "300 0.01" "30 0.01"; do set -- $cfg
  timeout -k 10 280 python3 tools/run_batched.py --rows 10000000 --batch 1024 --topn 100 --reps 5 --check 32 --catalogue clustered --clusters $1 --spread $2 \
    | sed "s/^{/{\"clusters\": $1, \"spread\": $2, /" >> $O/batched.jsonl
  timeout -k 10 280 python3 tools/run_half_multi.py --rows 10000000 --topn 100 --calls 20 --fp16 --sizes 12,32 --streams 12 --catalogue clustered --clusters $1 --spread $2 \
    | sed "s/^{/{\"clusters\": $1, \"spread\": $2, /" >> $O/multi.jsonl
done
cut -c1-700 $O/batched.jsonl
cut -c1-900 $O/multi.jsonl
