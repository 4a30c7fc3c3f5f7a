#!/bin/bash
# Round 6: the TN weight-gradient GEMM on the five-slot ring (BF_GEMM_TN_FORM=1, bf_gemm256_r5.hip TRX) against
# its two-buffer unit ring (0), developer library: back to back on the training step's shapes, then in the BERT-base training step.
OUT=$PWD/gpurun_out/r6h; mkdir -p $OUT; rm -f $OUT/tn_ring5_ab.txt
export BF_LIB_PATH=$PWD/bayeformers_amd/lib/libbayeformers_amd_dev.so
for round in 1 2; do for f in 0 1; do
  echo "round $round BF_GEMM_TN_FORM=$f" >> $OUT/tn_ring5_ab.txt
  BF_GEMM_TN_FORM=$f python3 tools/gemm_tn_bench.py 20 2048 768 768  10 4096 3072 768  10 4096 768 3072 2>/dev/null | sed 's/| nt.*//' >> $OUT/tn_ring5_ab.txt
done; done
for round in 1 2 3; do for f in 0 1; do
  BF_GEMM_TN_FORM=$f python3 bench.py --workload bert_base_train --no-cpu-baseline --no-traffic --steps 40 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round training step, BF_GEMM_TN_FORM=$f:', d['value'], 'MC-samples/s', d['ms_per_step'], 'ms/step', 'elbo', d['config']['last_elbo'])" >> $OUT/tn_ring5_ab.txt
done; done
cat $OUT/tn_ring5_ab.txt
