set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fz
for SEED in 7 1234567 606 20261005; do
FUZZ_SEED=$SEED FUZZ_CASES2=300 FUZZ_CASES=300 timeout -k 10 560 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/fz/fuzz_$SEED.log 2>&1 || { tail -60 gpurun_out/fz/fuzz_$SEED.log; exit 1; }
tail -2 gpurun_out/fz/fuzz_$SEED.log
done
