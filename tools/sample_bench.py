"""Where is the sampling kernel's time?  One Gaussian parameter of n scalars, S samples, MOPED-style Gaussian prior:
bf_sample_logprob with the sampled weights written (bf16) and without (log-probs only = the kernel's VALU work alone).

    python tools/sample_bench.py [n] [S]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bayeformers_amd.nn as bnn  # noqa: E402
from bayeformers_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 85_000_000 // 3072 * 3072
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    for prior_kind in ("gaussian", "mixture"):
        g = bnn.Gaussian(torch.Size((n // 3072, 3072))).cuda()
        if prior_kind == "gaussian":
            prior = bnn.Gaussian(torch.Size((n // 3072, 3072))).cuda()
            per = 16
        else:
            prior = bnn.DEFAULT_SCALED_GAUSSIAN_MIXTURE
            per = 8
        for out in (torch.bfloat16, None):
            ms = timeit(lambda: ops.sample_logprob([g], [prior], [0], S, 0x5EED, 0, out_dtype=out))
            byt = n * (per + (S * 2 if out is not None else 0))
            print(f"{prior_kind:8s} prior, n={n / 1e6:.1f} M, S={S}, out={'bf16' if out is not None else 'none'}: {ms:.3f} ms, "
                  f"{n * S / ms / 1e9:.2f} T eps/s, {byt / ms / 1e9:.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
