#!/bin/bash
# Round 6: per-position SQ and TCC counters of the tiled GEMM launches of the BERT-base step, on the
# binaries in the tree.  Two rocprofv3 --pmc passes (8 SQ slots + GRBM; 4 TCC slots), kernel-trace only, the program itself
# straight after `--`.   bash tools/r6a_pmc_positions.sh [tag] [extra env assignments for the bench, e.g. BF_LIB_PATH=...]
TAG=${1:-r6a}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "$@"; do export "$a"; done
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --graph off"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq -o t -- python3 $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum \
    --output-format csv -d $OUT/tcc -o t -- python3 $ARGS > $OUT/bench_tcc.json 2> $OUT/tcc.err
python3 tools/pmc_positions.py "PMC counters of the tiled GEMM launches by position (BERT-base S=10 B=32 L=128 bf16, eager steps; $TAG $*)" $OUT/sq $OUT/tcc > $OUT/pmc_gemm_positions.md
tail -5 $OUT/sq.err $OUT/tcc.err
cat $OUT/pmc_gemm_positions.md
find $OUT -name "*.csv" -size +20M -delete
