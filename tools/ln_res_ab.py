"""Developer A/B: add_layernorm with and without the residual stream, [S*B*L, 768] bf16 (the BERT-base step's shape)."""
import torch
import bayeformers_amd.ops as ops

rows, N = 10 * 32 * 128, 768
x = torch.randn(rows, N, device="cuda", dtype=torch.bfloat16)
r = torch.randn(rows, N, device="cuda", dtype=torch.bfloat16)
g = torch.ones(N, device="cuda")
b = torch.zeros(N, device="cuda")
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def run(res, flush):
    ts = []
    for _ in range(30):
        if flush:
            big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.add_layernorm(x, res, g, b, 1e-12)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for flush in (False, True):
    for _ in range(2):
        print(f"flush={flush} with residual {run(r, flush):.1f} us | without {run(None, flush):.1f} us")
