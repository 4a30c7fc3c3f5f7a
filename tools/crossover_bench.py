"""Mid-M crossover of the three ways to run one bnn.Linear forward:

    A  sampling launch + tiled GEMM          (bf_linear_fwd above the fused threshold: what large M runs)
    B  the single fused kernel               (bf_fused_small: epsilon in registers -> MFMA operand; M <= 128)
    C  the weight-stationary fused launch    (bf_linear_fwd_ws: a strip of sampled weights resident in LDS)

    python tools/crossover_bench.py  >  profiles/r4*_mid_m_crossover.txt

Sweeps M in {32, 64, 96, 128, 256, 512, 1024, 2048} x (N, K) in {512^2, 768^2, 3072x768} at S = 5 and 10, bf16, mixture and
MOPED priors.  Microseconds per call from HIP events around back-to-back calls, interleaved rounds, median.  All three
give the same outputs (checked)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bayeformers_amd as bf  # noqa: E402
import bayeformers_amd.nn as bnn  # noqa: E402
from bayeformers_amd import _C, ops  # noqa: E402


def timeit(fn, iters):
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
    lib = _C.lib()
    default_rows = lib.bf_fused_small_max_rows()
    print(f"# built-in fused threshold: {default_rows} rows per sample")
    bf.set_compute_dtype("bf16")
    shapes = [(512, 512), (768, 768), (3072, 768)]
    Ms = [32, 64, 96, 128, 256, 512, 1024, 2048]
    for prior in ("mixture", "moped"):
        for S in (5, 10):
            for N, K in shapes:
                torch.manual_seed(0)
                layer = bnn.Linear(K, N) if prior == "mixture" else bnn.Linear.from_frequentist(
                    torch.nn.Linear(K, N), delta=0.05, freeze=True)
                layer = layer.cuda()
                layer.layer_id = 0
                for M in Ms:
                    x = torch.randn(S * M, K, device="cuda").to(torch.bfloat16)
                    lp = torch.zeros(S, 2, dtype=torch.float64, device="cuda")
                    seed, base = 0x5EED, 0
                    runs = {}
                    _C.check(lib.bf_set_fused_small_max_rows(0), "set")
                    runs["A two-launch"] = lambda: ops.linear_forward(layer, x, S, seed, base, lp)
                    ya = runs["A two-launch"]()
                    if M <= 128:
                        def fused():
                            _C.check(lib.bf_set_fused_small_max_rows(128), "set")
                            y = ops.linear_forward(layer, x, S, seed, base, lp)
                            _C.check(lib.bf_set_fused_small_max_rows(0), "set")
                            return y
                        runs["B fused-small"] = fused
                    for sh in (1, 2):
                        runs[f"C ws x{sh}"] = (lambda sh=sh: ops.linear_forward_ws(layer, x, S, seed, base, lp, sh))
                    for name, fn in runs.items():
                        y = fn()
                        torch.cuda.synchronize()
                        d = (y.float() - ya.float()).abs().max().item()
                        assert d <= 2.0 ** -6 * ya.float().abs().max().item() + 1e-3, (name, d)
                    iters = 50 if M * N * K * S < 4e9 else 10
                    times = {k: [] for k in runs}
                    for _ in range(5):
                        for k, fn in runs.items():
                            times[k].append(timeit(fn, iters))
                    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
                    best = min(med, key=med.get)
                    flop = 2.0 * S * M * N * K
                    print(f"{prior:8s} S={S:2d} {N:4d}x{K:4d} M={M:4d} | " +
                          " | ".join(f"{k} {med[k]:7.1f} us" for k in runs) + f" | best: {best} ({flop / med[best] / 1e6:6.1f} TFLOP/s)",
                          flush=True)
    _C.check(lib.bf_set_fused_small_max_rows(default_rows), "set")


if __name__ == "__main__":
    main()
