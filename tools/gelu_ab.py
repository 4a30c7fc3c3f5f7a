"""Developer A/B of the FFN-up GEMM (S=10, M=4096, N=3072, K=768, GELU epilogue) — run once per library build:
BF_LIB_PATH=.../libbayeformers_amd_dev.so python tools/gelu_ab.py   (prints us per launch and the error against fp64 GELU)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402
from tools.gemm_tn_bench import timed  # noqa: E402

S, M, N, K = 10, 4096, 3072, 768
torch.manual_seed(0)
x = torch.randn(S, M, K, device="cuda").bfloat16()
w = (torch.randn(S, N, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(S, N, device="cuda")
t = timed(lambda: ops.gemm_nt(x, w, b, S, M, N, K, M * K, torch.bfloat16, 1), 10)
y = ops.gemm_nt(x[:1], w[:1], b[:1], 1, M, N, K, M * K, torch.bfloat16, 1).double()
pre = torch.einsum("smk,snk->smn", x[:1].double(), w[:1].double()) + b[:1, None, :].double()
ref = torch.nn.functional.gelu(pre)
err = (y - ref).abs()
print(f"{os.path.basename(os.environ.get('BF_LIB_PATH', 'product'))}: {t * 1e3:.1f} us  {2.0 * S * M * N * K / t / 1e9:.0f} TF | "
      f"max |err| {err.max().item():.3e}  max |err| / max|y| {err.max().item() / ref.abs().max().item():.3e}  "
      f"mean |err| {err.mean().item():.3e}")
