"""Why is the first GEMM after a LayerNorm slower than the next two on the same input?
Times three back-to-back 768x768 GEMMs (S=10, M=4096) after (a) a LayerNorm that rewrites their input x,
(b) a LayerNorm that writes somewhere else (x stays as the previous iteration left it), (c) nothing."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402

S, M, N, K = 10, 4096, 768, 768
x = torch.randn(S * M, K, device="cuda").bfloat16()
r = torch.randn(S * M, K, device="cuda").bfloat16()
other = torch.empty_like(x)
gam, bet = torch.ones(K, device="cuda").bfloat16(), torch.zeros(K, device="cuda").bfloat16()
ws = [torch.randn(S, N, K, device="cuda").bfloat16() for _ in range(3)]
b = torch.randn(S, N, device="cuda")
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
tiny = torch.zeros(64, device="cuda")
ma, mb = torch.randn(2048, 2048, device="cuda").bfloat16(), torch.randn(2048, 2048, device="cuda").bfloat16()
mc = torch.empty(2048, 2048, device="cuda", dtype=torch.bfloat16)


def run(mode, iters=40):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(iters)]
    for it in range(iters):
        if mode == "ln_rewrites_x":
            x.copy_(ops.add_layernorm(x, r, gam, bet, 1e-12))
        elif mode == "ln_inplace_out":  # LN output IS the GEMM input (as in the model)
            xin = ops.add_layernorm(x, r, gam, bet, 1e-12)
        elif mode == "ln_elsewhere":
            other.copy_(ops.add_layernorm(other, r, gam, bet, 1e-12))
        elif mode == "ln_then_tiny":
            xin = ops.add_layernorm(x, r, gam, bet, 1e-12)
            tiny.add_(1)
        elif mode == "ln_then_20us":
            xin = ops.add_layernorm(x, r, gam, bet, 1e-12)
            torch.mm(ma, mb, out=mc)
        elif mode == "flush":
            big.zero_()
        src = xin if mode in ("ln_inplace_out", "ln_then_tiny", "ln_then_20us") else x
        for j in range(3):
            ev[it][j].record()
            ops.gemm_nt(src, ws[j], b, S, M, N, K, M * K, torch.bfloat16)
        ev[it][3].record()
    torch.cuda.synchronize()
    t = [sum(ev[it][j].elapsed_time(ev[it][j + 1]) for it in range(5, iters)) / (iters - 5) * 1e3 for j in range(3)]
    print(f"{mode:16s}: GEMM1 {t[0]:6.1f} us  GEMM2 {t[1]:6.1f} us  GEMM3 {t[2]:6.1f} us")


for mode in ("none", "ln_rewrites_x", "ln_inplace_out", "ln_then_tiny", "ln_then_20us", "flush", "none"):
    run(mode)
