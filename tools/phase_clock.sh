# On the GPU box: build the library with the phase clock, run tools/phase_clock.py, leave nothing behind
# (gpurun boxes are thrown away; locally, rebuild with `python -c "import __graft_entry__ as g; g.build()"`).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q8
P=spotify_recommender_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_PHASE_CLOCK -o $P/libmi355rec.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip
: > gpurun_out/q8/phase.jsonl
for R in 1000000 10000000; do
  timeout -k 10 120 python tools/phase_clock.py --rows $R --topn ${TOPN:-10} >> gpurun_out/q8/phase.jsonl
done
cat gpurun_out/q8/phase.jsonl
