#!/bin/bash
# Unfiltered kernel trace of the BERT-base training step: per-step kernel counts and times, and the timeline of one step.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python3 bench.py --workload bert_base_train --steps 4 --warmup 2 --graph off --no-traffic --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r6f'
kt=glob.glob(out+'/prof/**/*kernel_trace.csv', recursive=True)[0]
ev=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))]
ev.sort()
starts=[i for i,e in enumerate(ev) if 'bf_sample_table_kernel' in e[2]]
a,b=starts[-2],starts[-1]
seg=ev[a:b]
cnt=collections.Counter(); tim=collections.Counter()
for s,e,n in seg:
    cnt[n]+=1; tim[n]+=(e-s)/1e3
with open(out+'/train_step_kernels.txt','w') as f:
    f.write(f"one training step: {len(seg)} kernels, span {(seg[-1][1]-seg[0][0])/1e3:.1f} us, busy {sum(tim.values()):.1f} us\n")
    for n,t in tim.most_common():
        f.write(f"{cnt[n]:5d} {t:9.1f} us  avg {t/cnt[n]:7.1f}  {n[:150]}\n")
t0=seg[0][0]
with open(out+'/train_step_timeline.txt','w') as f:
    for i,(s,e,n) in enumerate(seg):
        gap=(s-seg[i-1][1])/1e3 if i else 0
        f.write(f"{(s-t0)/1e3:9.1f} dur {(e-s)/1e3:7.1f} gap {gap:7.2f} {n[:110]}\n")
print(open(out+'/train_step_kernels.txt').read()[:7000])
PY
