#!/bin/bash
# Final measurements of a round, on the GPU box (gpurun):   bash tools/final_profiles.sh <round tag> <commit hash>
# Writes under gpurun_out/<tag>_final/: the bench lines of every workload, the rocprofv3 kernel trace of the default
# workload (raw kernel_stats.csv + kernel_trace.csv kept) and its summary stamped with the commit the binaries were built from.
# Copy the directory's *.json / *.md / *.csv into profiles/ afterwards (tools/final_profiles.sh does not touch profiles/).
set -u
TAG=${1:-rX}
HASH=${2:-unknown}
OUT=gpurun_out/${TAG}_final
mkdir -p $OUT
python bench.py > $OUT/bench_bert_base_default.json 2> $OUT/bench_bert_base_default.err
for w in bert_large_qa bert_base_serial bert_base_train bert_large_qa_train linear768 linear768_m32 mlp; do
    python bench.py --workload $w --no-traffic > $OUT/bench_$w.json 2> /dev/null
done
python bench.py --workload bert_base_train --no-dropout --no-traffic > $OUT/bench_bert_base_train_no_dropout.json 2> /dev/null
python bench.py --dtype fp32 --steps 5 --warmup 2 --no-traffic --no-cpu-baseline > $OUT/bench_bert_base_fp32.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bert -- python3 bench.py --steps 10 --warmup 3 --no-traffic --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /dev/null
python tools/rocprof_positions.py $OUT/prof/bert_kernel_trace.csv "Sources at commit $HASH (binaries built from it by python -m bayeformers_amd.build); command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-traffic --no-cpu-baseline; the bench line of this profiled run: bench_under_rocprof.json (profiled runs clock 2-3 % lower than unprofiled ones)." > $OUT/bert_base_final.md
cp $OUT/prof/bert_kernel_stats.csv $OUT/bert_base_final_kernel_stats.csv
# the trace itself is large: keep only the GEMM / sampling / LayerNorm / attention rows the summary is computed from
python - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/prof/bert_kernel_trace.csv")))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm256", "bf_sample", "add_layernorm", "attention_fwd", "fused_small", "embed_layernorm"))]
with open(f"{out}/bert_base_final_kernel_trace_path_kernels.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"], extrasaction="ignore")
    w.writeheader()
    w.writerows(keep)
PY
rm -rf $OUT/prof
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], "ms/step", "frac", r.get("frac"), "traffic", r.get("traffic"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
head -40 $OUT/bert_base_final.md | tail -14
