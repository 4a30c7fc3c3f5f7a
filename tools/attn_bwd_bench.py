"""Developer micro-benchmark of bf_attention_bwd at the BERT-base training shape (S*B = 320 sequences, T = 128,
12 heads) on the stacked q/k/v layout.   python tools/attn_bwd_bench.py [B T H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402
from tools.gemm_tn_bench import timed  # noqa: E402


def main():
    B, T, H = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (320, 128, 12)
    qkv = torch.randn(3, B, T, H * 64, device="cuda").bfloat16()
    q, k, v = (t.view(B, T, H, 64).transpose(1, 2) for t in qkv)
    out, lse = ops.attention_forward(q, k, v, None, 0.125, None, want_lse=True)
    go = torch.randn(B, T, H, 64, device="cuda").bfloat16()
    t_f = timed(lambda: ops.attention_forward(q, k, v, None, 0.125, None, want_lse=True), 10)
    t_b = timed(lambda: ops.attention_backward(q, k, v, None, None, out, go, lse, 0.125), 10)
    byts = B * T * H * 64 * 2
    print(f"B={B} T={T} H={H}: fwd {t_f * 1e3:6.1f} us ({4 * byts / t_f / 1e9:5.2f} TB/s of q,k,v,o)   "
          f"bwd {t_b * 1e3:6.1f} us ({8 * byts / t_b / 1e9:5.2f} TB/s of 5 reads + 3 writes)", flush=True)


if __name__ == "__main__":
    main()
