"""Yardstick: the vendor library (torch.bmm -> hipBLASLt/rocBLAS) on the sampled-weight GEMM's exact shapes,
next to bf_gemm_nt.  y[s] = x[s] @ W_s^T, bf16, S=10."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402


def t_us(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for S, M, N, K in [(10, 4096, 768, 768), (10, 4096, 3072, 768), (10, 4096, 768, 3072), (10, 6144, 1024, 1024),
                   (10, 6144, 4096, 1024), (10, 6144, 1024, 4096), (1, 4096, 4096, 4096)]:
    x = torch.randn(S, M, K, device="cuda").bfloat16()
    w = torch.randn(S, N, K, device="cuda").bfloat16()
    b = torch.randn(S, N, device="cuda")
    flops = 2.0 * S * M * N * K
    y = torch.empty(S, M, N, device="cuda", dtype=torch.bfloat16)
    wt = w.transpose(1, 2)
    print(f"S={S} M={M} N={N} K={K}", flush=True)
    def per_sample():
        for s_ in range(S):
            torch.nn.functional.linear(x[s_], w[s_], bb[s_])
    bb = b.bfloat16()
    tb = t_us(per_sample)
    print("  S x F.linear ok", flush=True)
    if os.environ.get("BF_TRY_BMM"):
        tv = t_us(lambda: torch.bmm(x, wt))
    else:
        tv = float("nan")
    to = t_us(lambda: ops.gemm_nt(x.view(S * M, K), w, b, S, M, N, K, M * K, torch.bfloat16))
    print(f"S={S} M={M} N={N} K={K}: torch.bmm {tv:7.1f} us = {flops / tv / 1e6:6.0f} TF | S x F.linear(+bias) {tb:7.1f} us = "
          f"{flops / tb / 1e6:6.0f} TF | bf_gemm_nt(+bias) {to:7.1f} us = {flops / to / 1e6:6.0f} TF")
