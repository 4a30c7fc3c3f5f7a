#!/bin/bash
# Round-5 A/B (one box, interleaved): does sampling each ~96 MB group right before the GEMMs that read it (ring of 1 or 2
# arenas: sampled weights possibly still in the 256 MB Infinity Cache) beat ONE whole-model sampling launch (default)?
OUT=gpurun_out/r5p; mkdir -p $OUT
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r.get('sample_kernel') or {}
print('$1', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'with_sampling', r.get('frac_with_sampling'), 'gemm_ms', r.get('gemm_ms_per_step'), 'sampling_ms', s.get('ms_per_step'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3; do
  for a in 17179869184 402653184 201326592 100663296; do
    BF_PLAN_ARENA_BYTES=$a python3 bench.py --steps 200 --warmup 5 --no-traffic --no-cpu-baseline 2>$OUT/err_$a.txt | line "round$round arena=$a" >> $OUT/arena_ring.txt
  done
done
cat $OUT/arena_ring.txt
