# On the GPU box: every route over CONTIGUOUS clusters (a catalogue sorted by genre / artist: spotify_recommender_amd/synth.py,
# clustered_catalogue(contiguous=True[, ramp=True])) at 10 M rows x top-100 — the single-query scans (fp32 rows, 8-bit
# replica: tools/run_replica.py), the multi-query pass (12 and 32 queries, a call alone and a stream: tools/run_half_multi.py)
# and the 1024-query batch (tools/run_batched.py) — plus the same tools over uniform rows for the "not worse" check.
# Keys are compared with the single-query fp32 scan inside each tool; the ORACLE comparison is tests/test_gpu_clustered.py.
#   bash tools/clustered.sh [out-dir]          (about 4 minutes)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/cl5}
mkdir -p $O
: > $O/single.jsonl
: > $O/batched.jsonl
: > $O/multi.jsonl
run() {   # $1 = tag json prefix, rest = catalogue flags
  tag=$1; shift
  timeout -k 10 280 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 200 --check 24 "$@" \
    | sed "s/^{/{$tag, /" >> $O/single.jsonl
  timeout -k 10 280 python3 tools/run_half_multi.py --rows 10000000 --topn 100 --calls 20 --fp16 --sizes 12,32 --streams 12,32 "$@" \
    | sed "s/^{/{$tag, /" >> $O/multi.jsonl
  timeout -k 10 280 python3 tools/run_batched.py --rows 10000000 --batch 1024 --topn 100 --reps 5 --check 32 "$@" \
    | sed "s/^{/{$tag, /" >> $O/batched.jsonl
}
run '"catalogue": "uniform"'
run '"catalogue": "contiguous", "clusters": 3000, "spread": 0.03, "ramp": false' --catalogue clustered --contiguous --clusters 3000 --spread 0.03
run '"catalogue": "contiguous", "clusters": 3000, "spread": 0.03, "ramp": true' --catalogue clustered --contiguous --ramp --clusters 3000 --spread 0.03
run '"catalogue": "contiguous", "clusters": 300, "spread": 0.01, "ramp": false' --catalogue clustered --contiguous --clusters 300 --spread 0.01
run '"catalogue": "contiguous", "clusters": 300, "spread": 0.01, "ramp": true' --catalogue clustered --contiguous --ramp --clusters 300 --spread 0.01
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
for name in ("single", "multi", "batched"):
    for line in open(f"{o}/{name}.jsonl"):
        d = json.loads(line)
        tag = f'{d.get("catalogue")} {d.get("clusters", "")} ramp={d.get("ramp", "")}'
        if name == "single":
            print(tag, "fp32", d["fp32_rows"]["us_per_step"], "q8", d["replica_q8"]["us_per_step"], d["replica_q8"]["rescored_per_query"],
                  "p50", d["fp32_rows"]["p50_us"], d["replica_q8"]["p50_us"], "same", d.get("keys_identical_q8"), d.get("streamed_identical_q8"))
        elif name == "multi":
            print(tag, "calls", [(c["queries"], c["pass_kernel_us"], c["rows_to_exact_chain_per_query"]) for c in d["single_calls"]],
                  "streams", [(c["queries"], c["us_per_call"], c["launch_kernel_us"]) for c in d["streams"]], "same", d.get("matches_single_query_fp32_scan"))
        else:
            print(tag, "ms", d["ms_per_batch"], "cand", d["candidates_total"], d["candidates_max"], "queued", d["queued_queries"], "same", d.get("matches_single_query_fp32_scan"))
PY
