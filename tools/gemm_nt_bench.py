"""Developer micro-benchmark of the forward GEMM (bf_gemm_nt_act, 16-bit output) at the BERT-base shapes.
python tools/gemm_nt_bench.py [S M N K] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402
from tools.gemm_tn_bench import timed  # noqa: E402


def main():
    v = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(v[i:i + 4]) for i in range(0, len(v), 4)] or [
        (10, 4096, 768, 768), (10, 4096, 2304, 768), (10, 4096, 3072, 768), (10, 4096, 768, 3072), (10, 4096, 3072, 3072)]
    for S, M, N, K in shapes:
        x = torch.randn(S, M, K, device="cuda").bfloat16()
        w = (torch.randn(S, N, K, device="cuda") * 0.05).bfloat16()
        b = torch.randn(S, N, device="cuda")
        flop = 2.0 * S * M * N * K
        t = timed(lambda: ops.gemm_nt(x, w, b, S, M, N, K, M * K, torch.bfloat16, 0), 10)
        print(f"S={S} M={M} N={N} K={K}: nt {t * 1e3:7.1f} us {flop / t / 1e9:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
