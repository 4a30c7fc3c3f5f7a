#!/bin/bash
# Round 6: tile-schedule policies A/B in the BERT-base step on ONE box, interleaved rounds, with the
# TCC counters of every arm beside the times.  Developer library (BF_GEMM_SCHED selects the policy at run time):
#   12 = round-5 default; 0x100c = + remainder tiles on adjacent columns; 0x300c = + column-group size chosen by the fetch model.
OUT=$PWD/gpurun_out/r6b; mkdir -p $OUT; rm -f $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BF_LIB_PATH=$PWD/bayeformers_amd/lib/libbayeformers_amd_dev.so
POLICIES="${POLICIES:-12 4108 12300}"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'samples/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'with_sampling', r.get('frac_with_sampling'), 'gemm_ms', r.get('gemm_ms_per_step'), 'elbo', d['config'].get('last_elbo'))"; }
for round in 1 2 3; do
  for pol in $POLICIES; do
    BF_GEMM_SCHED=$pol python3 bench.py --steps 100 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | line "round$round policy=$pol" >> $OUT/ab.txt
  done
done
for pol in $POLICIES; do
  export BF_GEMM_SCHED=$pol
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_$pol -o t -- \
      python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --graph off > /dev/null 2> $OUT/tcc_$pol.err
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_$pol -o t -- \
      python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --graph off > /dev/null 2> $OUT/sq_$pol.err
  python3 tools/pmc_positions.py "schedule policy $pol (developer library, BF_GEMM_SCHED=$pol): GEMM launches by position" $OUT/tcc_$pol $OUT/sq_$pol > $OUT/pmc_policy_$pol.md
done
unset BF_GEMM_SCHED
cat $OUT/ab.txt
for pol in $POLICIES; do grep -A8 "^## derived" $OUT/pmc_policy_$pol.md; done
find $OUT -name "*.csv" -size +8M -delete
