#!/bin/bash
# Round 6: where does the weight-gradient (TN) GEMM's time go?  The developer library's ablation switches
# (BF_GEMM_ABLATE) on the four TN shapes of the BERT-base training step, back to back (tools/gemm_tn_bench.py), against the
# forward (NT, fp32-out burst kernel) form of the same flop:
#   0 = the kernel as it ships; 16 = no global stores; 8 = no epilogue at all; 64 = every k-step's DMA re-reads k-step 0
#   (operands L2-hot); 1 = no DMA in the k-loop (fragment reads + MFMAs + barriers only); 9 = no DMA and no epilogue.
OUT=$PWD/gpurun_out/r6g; mkdir -p $OUT; rm -f $OUT/tn_probe.txt
export BF_LIB_PATH=$PWD/bayeformers_amd/lib/libbayeformers_amd_dev.so
for ab in 0 16 8 64 1 9 0; do
  echo "BF_GEMM_ABLATE=$ab" >> $OUT/tn_probe.txt
  BF_GEMM_ABLATE=$ab python3 tools/gemm_tn_bench.py 20 2048 768 768  10 4096 3072 768  10 4096 768 3072 2>/dev/null | sed 's/| nt.*//' >> $OUT/tn_probe.txt
done
cat $OUT/tn_probe.txt
