mkdir -p gpurun_out
python bench.py 2>gpurun_out/r2d_default.err > gpurun_out/r2d_bench_bert_base_default.json; tail -c 2500 gpurun_out/r2d_bench_bert_base_default.json
for w in bert_large_qa linear768 linear768_m32 mlp bert_base_train; do
  python bench.py --workload $w --no-traffic 2>/dev/null > gpurun_out/r2d_bench_$w.json; python -c "
import json,sys; d=json.loads(open('gpurun_out/r2d_bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], r['kernel'][:40], r['achieved'], r['frac'], r.get('sample_kernel',{}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))"
done
