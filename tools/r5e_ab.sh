#!/bin/bash
# Round-5 A/B session (one box, interleaved rounds): VERDICT r4 item 3 variants against the product library.
#   ea2 / ea4  bf_gemm256_r5.hip -DBF_R5_EARLY_ACT=1 (-DBF_R5_EARLY_LAG=4): FFN-up's GELU under the last MFMA slot (3b)
#   pair       bf_sample.hip -DBF_SAMPLE_PAIR=2: two table blocks per workgroup at S <= 2, second block's loads in flight (3d)
#   at1 / at3  bf_attention.hip -DBF_ATTN_OUT_STORES=1|3: nontemporal / write-through attention output rows
# and, on the developer library, tiles of height 4 (128 x 256, what a two-accumulator-set variant would run: 3a) with and
# without their epilogue against the product's height-8 tiles.
OUT=gpurun_out/r5e; mkdir -p $OUT
L=$PWD/bayeformers_amd/lib
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r.get('sample_kernel') or {}
print('$1', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'with_sampling', r.get('frac_with_sampling'), 'gemm_ms', r.get('gemm_ms_per_step'), 'sampling_ms', s.get('ms_per_step'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3; do
  for v in base ea2 ea4 at1 at3; do
    lib=$L/libbayeformers_amd.so; [ $v != base ] && lib=$L/libbayeformers_amd_$v.so
    BF_LIB_PATH=$lib python3 bench.py --steps 300 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round $v" >> $OUT/gemm_variants_in_step.txt
  done
done
for round in 1 2 3; do
  for S in 1 2; do
    for v in base pair; do
      lib=$L/libbayeformers_amd.so; [ $v != base ] && lib=$L/libbayeformers_amd_$v.so
      BF_LIB_PATH=$lib python3 bench.py --samples $S --steps 400 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round S=$S $v" >> $OUT/sampling_pair_small_shards.txt
    done
  done
done
# tile height 4 vs 8 on the ring kernel (developer library): variant:policy:ablate:form — policy 12 | hmax << 4; ablate 8 = no epilogue
export LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH
for shape in "10 4096 768 768 0" "10 4096 2304 768 0" "10 4096 3072 768 1" "10 4096 768 3072 0"; do
  echo "== S M N K act = $shape" >> $OUT/tile_height_4_vs_8.txt
  BF_BENCH_CONFIGS="2:12:0:2 2:76:0:2 2:12:8:2 2:76:8:2" tools/bin/gemm_bench $shape >> $OUT/tile_height_4_vs_8.txt 2>&1
done
tail -n 40 $OUT/*.txt
