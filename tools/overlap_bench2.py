"""Does the VALU-bound sampling kernel hide under the wrapped model's memory-bound kernels (flash attention,
residual+LayerNorm) when issued on a second HIP stream?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bayeformers_amd as bf, bayeformers_amd.nn as bnn
from bayeformers_amd import ops

torch.manual_seed(0)
S = 10
lin = torch.nn.Linear(768, 3072)
lb = bnn.Linear.from_frequentist(lin, delta=0.05, freeze=True).cuda()
q = torch.randn(320, 12, 128, 64, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
xs = torch.randn(40960, 768, device="cuda").bfloat16(); rs = torch.randn_like(xs)
g, b = torch.ones(768, device="cuda").bfloat16(), torch.zeros(768, device="cuda").bfloat16()
side = torch.cuda.Stream()

def attn(): return torch.nn.functional.scaled_dot_product_attention(q, k, v)
def ln(): return ops.add_layernorm(xs, rs, g, b, 1e-12)
def sample(): return ops.sample_logprob([lb.weight], [lb.weight_prior], [2], S, 1, 0, out_dtype=torch.bfloat16)

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

def overlapped(main):
    def f():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            sample()
            done = torch.cuda.Event(); done.record()
        main()
        torch.cuda.current_stream().wait_event(done)
    return f

ts = timeit(sample)
for name, fn in (("attention", attn), ("add+layernorm", ln), ("3 x attention", lambda: (attn(), attn(), attn()))):
    tm = timeit(fn)
    to = timeit(overlapped(fn))
    print(f"{name:14s}: alone {tm:6.1f} us, sampling alone {ts:6.1f} us, serial {tm + ts:6.1f} us, two streams {to:6.1f} us")
