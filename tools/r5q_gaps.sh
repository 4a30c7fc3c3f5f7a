#!/bin/bash
# What sits in the ~4.7 us gaps in front of three of the four GEMM launches of a layer?  Unfiltered kernel + memory-copy trace.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python3 bench.py ${1:+--workload $1} --steps 4 --warmup 2 --no-traffic --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import csv, glob, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r5q'
kt=glob.glob(out+'/prof/**/*kernel_trace.csv', recursive=True)[0]
ev=[]
for r in csv.DictReader(open(kt)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K '+r['Kernel_Name'][:90]))
mc=glob.glob(out+'/prof/**/*memory_copy_trace.csv', recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C '+r.get('Direction','')+' '+r.get('Bytes', r.get('Size',''))))
ev.sort()
starts=[i for i,e in enumerate(ev) if 'bf_sample_table_kernel' in e[2]]
a,b=starts[-2],starts[-1]
t0=ev[a][0]
with open(out+'/one_step_timeline.txt','w') as f:
    for i in range(a,b):
        e=ev[i]; gap=(e[0]-ev[i-1][1])/1e3
        f.write(f"{(e[0]-t0)/1e3:9.1f} dur {(e[1]-e[0])/1e3:7.1f} gap {gap:7.2f} {e[2]}\n")
import collections
L=open(out+"/one_step_timeline.txt").read().splitlines()
print("\n".join(L[:14])); print("..."); print("\n".join(L[-40:]))
PY
