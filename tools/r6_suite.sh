# Round 6: every GPU test + smoke, then queries by value alone (latency), then PART A of the evidence runs.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6suite
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
for V in "" "--by-value"; do
  timeout -k 10 200 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300 --check 4 --only 2 $V 2>> $O/err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['by_value'], d['replica_q8'])"
  timeout -k 10 200 python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300 --check 4 --only 2 --catalogue clustered --contiguous --ramp --clusters 3000 --spread 0.03 $V 2>> $O/err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['by_value'], d['replica_q8'])"
done
PART=A bash tools/collect_r06.sh
