#!/bin/bash
# Round 6: (1) the FFN pair as one autograd node (GELU' in the down-projection's input-gradient GEMM) on / off in the training
# step; (2) the round-5 tree (tools/_r5tree = a worktree of aaeba75, built there) against this one, same box, interleaved:
# the default forward workload and the training step.
OUT=$PWD/gpurun_out/r6i; mkdir -p $OUT; rm -f $OUT/ab.txt
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], 'MC-samples/s', d['ms_per_step'], 'ms/step', 'frac', r.get('frac'), 'fws', r.get('frac_with_sampling'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3; do
  BF_BENCH_NO_FFN_PAIR=1 python3 bench.py --workload bert_base_train --no-cpu-baseline --no-traffic --steps 40 --warmup 4 2>/dev/null | line "round$round train, two autograd nodes  " >> $OUT/ab.txt
  python3 bench.py --workload bert_base_train --no-cpu-baseline --no-traffic --steps 40 --warmup 4 2>/dev/null | line "round$round train, FFN pair one node   " >> $OUT/ab.txt
done
if [ -d tools/_r5tree ]; then
for round in 1 2 3; do
  (cd tools/_r5tree && python3 bench.py --workload bert_base_train --no-cpu-baseline --no-traffic --steps 40 --warmup 4 2>/dev/null) | line "round$round train   round-5 tree (aaeba75)" >> $OUT/ab.txt
  python3 bench.py --workload bert_base_train --no-cpu-baseline --no-traffic --steps 40 --warmup 4 2>/dev/null | line "round$round train   this tree            " >> $OUT/ab.txt
  (cd tools/_r5tree && python3 bench.py --no-cpu-baseline --no-traffic --steps 100 --warmup 5 2>/dev/null) | line "round$round forward round-5 tree (aaeba75)" >> $OUT/ab.txt
  python3 bench.py --no-cpu-baseline --no-traffic --steps 100 --warmup 5 2>/dev/null | line "round$round forward this tree            " >> $OUT/ab.txt
done
fi
cat $OUT/ab.txt
