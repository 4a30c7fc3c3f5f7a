"""Developer micro-benchmark of the weight-gradient GEMM (bf_gemm_tn) at the BERT-base training shapes, next to the
forward GEMM (bf_gemm_nt_act) of the same FLOPs.   python tools/gemm_tn_bench.py [batch Mc N K] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402


def timed(fn, iters):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def main():
    v = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(v[i:i + 4]) for i in range(0, len(v), 4)] or [
        (20, 2048, 768, 768), (10, 4096, 768, 768), (10, 4096, 3072, 768), (10, 4096, 768, 3072), (40, 1024, 768, 768)]
    for batch, Mc, N, K in shapes:
        a = torch.randn(batch, Mc, N, device="cuda").bfloat16()
        b = torch.randn(batch, Mc, K, device="cuda").bfloat16()
        flop = 2.0 * batch * Mc * N * K
        t_tn = timed(lambda: ops.gemm_tn(a, b), 10)
        # the NT kernel on the same FLOPs: x [batch][N][Mc], w [batch][K][Mc]
        at, bt = a.transpose(1, 2).contiguous(), b.transpose(1, 2).contiguous()
        t_nt = timed(lambda: ops.gemm_nt(at, bt, None, batch, N, K, Mc, N * Mc, torch.float32), 10)
        print(f"batch={batch} Mc={Mc} N={N} K={K}: tn {t_tn * 1e3:7.1f} us {flop / t_tn / 1e9:6.0f} TF | "
              f"nt {t_nt * 1e3:7.1f} us {flop / t_nt / 1e9:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
