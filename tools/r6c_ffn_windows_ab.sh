#!/bin/bash
# Round 6: the FFN pair issued per window of the Monte-Carlo sample axis (fuse_ffn_pairs: FFN-up then
# FFN-down for half of the samples, then for the other half) so that FFN-down reads its 126 MB x from the Infinity Cache.
# One box, interleaved rounds; then the full bench line (traffic legs: infinity_cache_hit_fraction by position) per arm.
OUT=$PWD/gpurun_out/r6c; mkdir -p $OUT; rm -f $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WINDOWS="${WINDOWS:-1 2 5}"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'with_sampling', r.get('frac_with_sampling'), 'gemm_ms', r.get('gemm_ms_per_step'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3; do
  for w in $WINDOWS; do
    BF_BENCH_FFN_WINDOWS=$w python3 bench.py --steps 100 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round windows=$w" >> $OUT/ab.txt
  done
done
for w in 1 2; do
  BF_BENCH_FFN_WINDOWS=$w python3 bench.py --no-cpu-baseline > $OUT/bench_windows_$w.json 2> $OUT/bench_windows_$w.err
  python3 -c "
import json
d=json.loads(open('$OUT/bench_windows_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print('windows=$w', d['value'], d['ms_per_step'], r['frac'], r['frac_with_sampling'], 'traffic', r.get('traffic'))
for k,v in (r.get('traffic_detail') or {}).get('by_position', {}).items(): print('   ', k, v)
" >> $OUT/ab.txt
done
cat $OUT/ab.txt
