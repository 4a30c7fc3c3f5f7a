#!/bin/bash
# Round-5 A/B: attention forward — old = the kernel before this change (per-block mask branches, ds_bpermute reductions, scale as
# its own fma), base = mask-specialised tile bodies + v_permlane swaps + the scale inside the exponent's fma.
OUT=gpurun_out/r5t; mkdir -p $OUT; rm -f $OUT/attn_ab.txt
L=$PWD/bayeformers_amd/lib
for round in 1 2 3; do
  for v in old base; do
    lib=$L/libbayeformers_amd.so; [ $v = old ] && lib=$L/libbayeformers_amd_attnold.so
    for shape in "160 16 384 fp16" "320 12 128"; do
      echo -n "round$round $v " >> $OUT/attn_ab.txt
      BF_LIB_PATH=$lib python3 tools/attn_fwd_bench.py $shape 2>/dev/null | head -1 >> $OUT/attn_ab.txt
    done
  done
done
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'gemm_ms', r.get('gemm_ms_per_step'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3; do
  for v in old base; do
    lib=$L/libbayeformers_amd.so; [ $v = old ] && lib=$L/libbayeformers_amd_attnold.so
    BF_LIB_PATH=$lib python3 bench.py --steps 200 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round $v bert_base" >> $OUT/attn_ab.txt
    BF_LIB_PATH=$lib python3 bench.py --workload bert_large_qa --steps 40 --warmup 3 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round $v bert_large_qa" >> $OUT/attn_ab.txt
  done
done
cat $OUT/attn_ab.txt
