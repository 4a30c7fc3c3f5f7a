"""NS-1, settled by measurement: ONE fused weight-stationary launch (bf_linear_fwd_ws: the sampled weights never
touch HBM) against the shipped path (bf_linear_fwd: sampling + log-prob launch, then the 256-wide GEMM) on the
headline layer — bnn.Linear 768 -> 768, x = [4096, 768] per sample, S = 10, bf16.

    python tools/fused_ws_bench.py [S M N K]

Prints microseconds per call (HIP events around back-to-back calls, interleaved rounds) and checks that both give the
same outputs and log-probs.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bayeformers_amd as bf  # noqa: E402
import bayeformers_amd.nn as bnn  # noqa: E402
from bayeformers_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    S, M, N, K = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (10, 4096, 768, 768)
    torch.manual_seed(0)
    bf.set_compute_dtype("bf16")
    for prior in ("mixture", "moped"):
        if prior == "mixture":
            layer = bnn.Linear(K, N)
        else:
            layer = bnn.Linear.from_frequentist(torch.nn.Linear(K, N), delta=0.05, freeze=True)
        layer = layer.cuda()
        layer.layer_id = 0
        x = torch.randn(S * M, K, device="cuda").to(torch.bfloat16)
        lp_a = torch.zeros(S, 2, dtype=torch.float64, device="cuda")
        lp_b = torch.zeros_like(lp_a)
        seed, base = 0x5EED, 0
        ya = ops.linear_forward(layer, x, S, seed, base, lp_a)
        res = {}
        for shares in (1, 2, 4, 0):
            yb = ops.linear_forward_ws(layer, x, S, seed, base, lp_b, shares)
            torch.cuda.synchronize()
            dy = (ya.float() - yb.float()).abs().max().item()
            dlp = ((lp_a - lp_b).abs() / lp_a.abs()).max().item()
            res[shares] = (dy, dlp)
        rounds = []
        for _ in range(5):
            t_two = timeit(lambda: ops.linear_forward(layer, x, S, seed, base, lp_a))
            t_ws = {sh: timeit(lambda: ops.linear_forward_ws(layer, x, S, seed, base, lp_b, sh)) for sh in (1, 2, 4)}
            rounds.append((t_two, t_ws))
        med = lambda v: sorted(v)[len(v) // 2]
        flop = 2.0 * S * M * N * K
        t2 = med([r[0] for r in rounds])
        print(f"{prior:8s} S={S} M={M} N={N} K={K}: sampling launch + 256-wide GEMM {t2:7.1f} us ({flop / t2 / 1e6:6.0f} TFLOP/s)")
        for sh in (1, 2, 4):
            t = med([r[1][sh] for r in rounds])
            print(f"{'':8s}   fused weight-stationary, {sh} row share(s): {t:7.1f} us ({flop / t / 1e6:6.0f} TFLOP/s)  "
                  f"x{t / t2:.2f}   max|dy| {res[sh][0]:.3g}  max rel dlogprob {res[sh][1]:.2e}")


if __name__ == "__main__":
    main()
