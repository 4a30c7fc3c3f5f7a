#!/bin/bash
# Round-5 A/B (one box, interleaved): the ring GEMM with an L2 prefetch running 1 / 2 / 3 k-steps ahead of the LDS-DMA (-DBF_R5_L2PF).
OUT=gpurun_out/r5v; mkdir -p $OUT; rm -f $OUT/l2pf.txt
L=$PWD/bayeformers_amd/lib
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'with_sampling', r.get('frac_with_sampling'), 'gemm_ms', r.get('gemm_ms_per_step'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3; do
  for v in base l2pf1 l2pf2 l2pf3; do
    lib=$L/libbayeformers_amd.so; [ $v != base ] && lib=$L/libbayeformers_amd_$v.so
    BF_LIB_PATH=$lib python3 bench.py --steps 200 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round $v" >> $OUT/l2pf.txt
  done
done
cat $OUT/l2pf.txt
