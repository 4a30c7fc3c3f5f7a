#!/bin/bash
# Round 6: the new parity rows (c4 per shard, c5 samples 3 / 6, the reference's serial caller loop on BERT-base) with their printed
# errors, the whole GPU suite on the cleaned kernels, and the serial-loop workload timed eager vs replayed.
OUT=$PWD/gpurun_out/r6d; mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q -s -k "c4 or c5 or serial_loop or model_call_replays or dropped_model" > $OUT/pytest_new_parity.txt 2>&1
tail -3 $OUT/pytest_new_parity.txt
grep "^\[c" $OUT/pytest_new_parity.txt
python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
python3 bench.py --workload bert_base_serial --steps 20 --warmup 3 > $OUT/bench_bert_base_serial.json 2> $OUT/serial.err
BF_NO_AUTO_GRAPH=1 python3 bench.py --workload bert_base_serial --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_bert_base_serial_eager.json 2>> $OUT/serial.err
python3 bench.py --no-cpu-baseline --no-traffic --steps 50 > $OUT/bench_bert_base.json 2>> $OUT/serial.err
for f in bench_bert_base_serial bench_bert_base_serial_eager bench_bert_base; do python3 -c "
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'frac', r['frac'], r['frac_with_sampling'], 'graph', d['config']['hip_graph'])"; done
tail -3 $OUT/serial.err
