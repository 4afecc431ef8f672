# On the GPU box: build an INSTRUMENTED copy of the library (a clock store per workgroup and phase) under
# gpurun_out/ and run tools/phase_clock.py against it.  The product's libmi355rec.so is not touched, so nothing
# run afterwards in the same checkout measures or tests the instrumented build.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q8
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o gpurun_out/q8/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_PHASE_CLOCK ${EXTRA_DEFS} -o gpurun_out/q8/libmi355rec_phase.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,gpurun_out/q8/cpu_backend.o -lgomp
: > gpurun_out/q8/phase.jsonl
for R in 1000000 10000000; do
  timeout -k 10 120 python tools/phase_clock.py --lib gpurun_out/q8/libmi355rec_phase.so --rows $R --topn ${TOPN:-10} >> gpurun_out/q8/phase.jsonl
done
cat gpurun_out/q8/phase.jsonl
