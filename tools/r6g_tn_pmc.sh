#!/bin/bash
# Round 6: SQ counters of the TN weight-gradient kernel and, beside it, the forward ring kernel on the same flop (tools/gemm_tn_bench.py).
OUT=$PWD/gpurun_out/r6g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc -o t -- python3 tools/gemm_tn_bench.py 10 4096 3072 768 10 4096 768 3072 > /dev/null 2> $OUT/pmc.err
python3 tools/pmc_summary.py $OUT/pmc/t_counter_collection.csv gemm256 > $OUT/tn_pmc.md
cat $OUT/tn_pmc.md
