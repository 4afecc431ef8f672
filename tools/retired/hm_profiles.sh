# The multi-query pass's own evidence (fp16 front end = the route from three queries per pass up; the 8-bit front end beside it).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
python3 tools/run_half_multi.py --fp16 > $O/half_multi.json 2> $O/half_multi.err
python3 tools/run_half_multi.py > $O/half_multi_q8.json 2>> $O/half_multi.err
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_hm -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/trace_hm.log 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_hm_a -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_hm_b -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_hm_f -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_hm_w -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60 > $O/pmc_hm_w.log 2>&1
echo done
