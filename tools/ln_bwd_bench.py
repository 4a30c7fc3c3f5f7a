"""Residual + LayerNorm backward at the BERT-base training shape (S=10, B=32, L=128, N=768, bf16): with / without dropout and a second
consumer's gradient (twin):  python tools/ln_bwd_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402

S, rows, N = 10, 10 * 32 * 128, 768
g = torch.Generator(device="cuda").manual_seed(0)
x, res, go, go2 = (torch.randn(rows, N, device="cuda", generator=g).bfloat16() for _ in range(4))
gamma = torch.randn(N, device="cuda", generator=g)
drop = ops.Dropout(0.1, 1, 2, 3)


def timed(dr, tw):
    call = lambda: ops.add_layernorm_backward(x, res, gamma, go, 1e-12, drop if dr else None, grad_out2=go2 if tw else None)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


for r in range(2):
    for dr in (0, 1):
        for tw in (0, 1):
            mb = 63 * (4 + dr + tw)
            a = timed(dr, tw)
            print(f"round {r} dropout={dr} twin={tw}: {a:6.1f} us ({mb / a:.2f} TB/s)")
