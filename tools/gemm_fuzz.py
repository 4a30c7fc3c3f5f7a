"""Random-shape check of the three forms of the 256-wide GEMM (NT with bias / GELU / pre-activation output, TN, NN)
against fp64 einsums.   python tools/gemm_fuzz.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402


def run(n=60, seed=0):
    """Returns the number of failing cases (tests/test_gpu_linear.py::test_gemm_fuzz_all_forms_against_fp64 runs 65)."""
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n):
        S = int(rng.integers(1, 5))
        M = int(rng.integers(130, 3000))
        N = int(rng.integers(1, 140)) * 8
        K = int(rng.integers(1, 9)) * 64
        dt = torch.bfloat16 if rng.integers(0, 2) else torch.float16
        act = int(rng.integers(0, 2))
        g = torch.Generator(device="cuda").manual_seed(case)
        x = torch.randn(S, M, K, device="cuda", generator=g).to(dt)
        w = (torch.randn(S, N, K, device="cuda", generator=g) * 0.1).to(dt)
        b = torch.randn(S, N, device="cuda", generator=g)
        pre_ref = torch.einsum("smk,snk->smn", x.double(), w.double()) + b[:, None, :].double()
        tol = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        y = ops.gemm_nt(x, w, b, S, M, N, K, M * K, dt, act)
        ref = torch.nn.functional.gelu(pre_ref) if act else pre_ref
        e1 = (y.double() - ref).abs().max().item() / (tol * ref.abs().max().item() + 1e-5 * K ** 0.5)
        y2, pre = ops.gemm_nt_act_pre(x, w, b, S, M, N, K, M * K, dt, 1)
        e2 = (pre.double() - pre_ref).abs().max().item() / (tol * pre_ref.abs().max().item() + 1e-5 * K ** 0.5)
        e3 = (y2.double() - torch.nn.functional.gelu(pre.double())).abs().max().item() / (tol * pre_ref.abs().max().item() + 1e-6)
        # TN: contraction over rows (multiple of 64), NN: contraction over N (multiple of 64)
        Mc = (M // 64) * 64
        a = torch.randn(S, Mc, N, device="cuda", generator=g).to(dt)
        xt = x[:, :Mc].contiguous()
        tn = ops.gemm_tn(a, xt)
        rtn = torch.einsum("bmn,bmk->bnk", a.double(), xt.double())
        e4 = (tn.double() - rtn).abs().max().item() / (2e-6 * Mc ** 0.5 * rtn.abs().max().item() + 1e-9)
        e5 = 0.0
        if N % 64 == 0 and M * K >= 128 * 128:
            wn = (torch.randn(S, N, K, device="cuda", generator=g) * 0.1).to(dt)
            dy = torch.randn(S, M, N, device="cuda", generator=g).to(dt)
            nn_ = ops.gemm_nn(dy, wn)
            rnn = torch.einsum("smn,snk->smk", dy.double(), wn.double())
            e5 = (nn_.double() - rnn).abs().max().item() / (tol * rnn.abs().max().item() + 1e-5 * N ** 0.5)
        worst = max(e1, e2, e3, e4, e5)
        flag = "" if worst <= 1.0 else "  <-- FAIL"
        bad += worst > 1.0
        print(f"case {case}: S={S} M={M} N={N} K={K} {str(dt)[6:]} act={act}: err/tol nt {e1:.2f} pre {e2:.2f} act {e3:.2f} tn {e4:.2f} nn {e5:.2f}{flag}", flush=True)
    print("FAILED" if bad else "ok", bad)
    return bad


def main():
    return 1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0


if __name__ == "__main__":
    sys.exit(main())
