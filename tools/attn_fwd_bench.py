"""Developer micro-benchmark of bf_attention_fwd at the BERT-base shape (320 sequences x 12 heads x 128 tokens; or
`B H T [fp16]` on the command line, e.g. 160 16 384 fp16 = BERT-large QA), back to back and after a cache flush; with the
developer library BF_ATTN_ABLATE=1 drops the output stores, 2 stages K / V once from tile 0.
    BF_LIB_PATH=bayeformers_amd/lib/libbayeformers_amd_dev.so python tools/attn_fwd_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeformers_amd import ops  # noqa: E402


def main():
    B, H, T, D = 320, 12, 128, 64
    a = sys.argv[1:]
    if len(a) >= 3:
        B, H, T = int(a[0]), int(a[1]), int(a[2])
    dt = torch.float16 if "fp16" in a else torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(3, B, T, H * D, device="cuda", generator=g).to(dt)
    q, k, v = (qkv[i].view(B, T, H, D).transpose(1, 2) for i in range(3))
    big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for abl in os.environ.get("BF_ATTN_ABLATES", "0 1").split():
        os.environ["BF_ATTN_ABLATE"] = abl
        for _ in range(3):
            ops.attention_forward(q, k, v, None, D ** -0.5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hot, cold = [], []
        for _ in range(7):
            e0.record()
            for _ in range(10):
                ops.attention_forward(q, k, v, None, D ** -0.5)
            e1.record()
            e1.synchronize()
            hot.append(e0.elapsed_time(e1) * 100)
        for _ in range(9):
            big.fill_(1)
            e0.record()
            ops.attention_forward(q, k, v, None, D ** -0.5)
            e1.record()
            e1.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        hot.sort(); cold.sort()
        print(f"B={B} H={H} T={T} {dt}: BF_ATTN_ABLATE={abl}: back to back {hot[len(hot) // 2]:.1f} us, after a flush {cold[len(cold) // 2]:.1f} us (incl. ~5 us launch)")


if __name__ == "__main__":
    main()
