# On the GPU box: the batched two-pass path with and without tile skipping (batched.hip.h, kTileMax), pass 1 looking at
# every 2nd / 4th / 8th tile, at 10 M and 12.5 M rows x 1024 queries x top-100.  Uses an MI355REC_EXPERIMENTS build
# under gpurun_out/ (the step is an environment knob there); the product library is not touched.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bq
mkdir -p $O
P=spotify_recommender_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -I$P/csrc -ldl \
  -DMI355REC_EXPERIMENTS -o $O/libmi355rec_exp.so $P/csrc/mi355rec.hip $P/csrc/sharded.hip
: > $O/ab.jsonl
for R in ${ROWS:-10000000 12500000}; do
  for S in ${STEPS:-4 2 8}; do
    for PATHV in 2 5; do
      if [ $PATHV = 5 ] && [ $S != 4 ]; then continue; fi
      echo "rows $R step $S path $PATHV" >&2
      MI355REC_BQ_STEP1=$S timeout -k 10 120 python tools/run_batched.py --lib $O/libmi355rec_exp.so --rows $R --batch 1024 --reps 20 --path $PATHV | sed "s/^{/{\"step1\": $S, /" >> $O/ab.jsonl
    done
  done
done
cat $O/ab.jsonl
