# On the GPU box: an instrumented copy of the engine library (-DMI355REC_PHASE_CLOCK [-DMI355REC_EXPERIMENTS] ...) under the
# given directory — never the product library.  The tools that read the stamps take it with --lib.
#   bash tools/phase_build.sh gpurun_out/ph [extra hipcc flags]   ->  gpurun_out/ph/libmi355rec_phase.so
set -e
O=$1; shift
mkdir -p $O
P=spotify_recommender_amd
g++ -std=c++17 -O3 -fopenmp -ffp-contract=off -fPIC -Iinclude -I$P/csrc -c $P/csrc/cpu_backend.cpp -o $O/cpu_backend.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_PHASE_CLOCK "$@" -o $O/libmi355rec_phase.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip -Wl,$O/cpu_backend.o -lgomp
