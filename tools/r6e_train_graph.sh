#!/bin/bash
# Round 6: the training step eager vs replayed from a HIP graph (training.GraphedTrainingStep): host time per
# step and step time, BERT-base, S = 10 / 2 / 1 samples per step.
OUT=$PWD/gpurun_out/r6e; mkdir -p $OUT; rm -f $OUT/train_graph.txt
for S in 10 2 1; do
  for g in eager graph; do
    python3 tools/host_time.py bert_base_train $g $S 2>/dev/null | tail -1 >> $OUT/train_graph.txt
  done
done
for round in 1 2; do for g in off auto; do
  python3 bench.py --workload bert_base_train --graph $g --no-cpu-baseline --no-traffic --steps 40 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round$round bench bert_base_train --graph $g:', d['value'], 'MC-samples/s', d['ms_per_step'], 'ms/step', d['config']['hip_graph'])" >> $OUT/train_graph.txt
done; done
cat $OUT/train_graph.txt
