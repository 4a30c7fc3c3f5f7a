#!/bin/bash
# Round-5 A/B (one box, interleaved): the BERT-base step with and without the per-layer copy of the device counter
# (BF_AB_OLD_SNAPSHOT=1 restores one clone per Bayesian layer and forward, as before this change).
OUT=gpurun_out/r5r; mkdir -p $OUT
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r.get('sample_kernel') or {}
print('$1', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'with_sampling', r.get('frac_with_sampling'), 'gemm_ms', r.get('gemm_ms_per_step'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3 4; do
  for v in 1 0; do
    BF_AB_OLD_SNAPSHOT=$v python3 bench.py --steps 200 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round old_snapshot=$v" >> $OUT/counter_copy_ab.txt
  done
done
cat $OUT/counter_copy_ab.txt
