"""Does the VALU-bound sampling kernel overlap with the MFMA-bound GEMM when issued on two HIP streams?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bayeformers_amd as bf, bayeformers_amd.nn as bnn
from bayeformers_amd import ops

torch.manual_seed(0)
S, M = 10, 4096
def mk(N, K):
    lin = torch.nn.Linear(K, N)
    l = bnn.Linear.from_frequentist(lin, delta=0.05, freeze=True).cuda()
    return l
la, lb = mk(3072, 768), mk(3072, 768)
x = torch.randn(S * M, 768, device="cuda").bfloat16()
wa = ops.sample_logprob([la.weight], [la.weight_prior], [0], S, 1, 0, out_dtype=torch.bfloat16)[0][0]
side = torch.cuda.Stream()

def gemm():
    return ops.gemm_nt(x, wa, None, S, M, 3072, 768, M * 768, torch.bfloat16)
def sample():
    return ops.sample_logprob([lb.weight], [lb.weight_prior], [2], S, 1, 0, out_dtype=torch.bfloat16)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

def serial():
    gemm(); sample()
def overlapped():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        sample()
        done = torch.cuda.Event(); done.record()
    gemm()
    torch.cuda.current_stream().wait_event(done)

print("variant", os.environ.get("BF_GEMM_VARIANT", "default"))
print("gemm only   %.1f us" % timeit(gemm))
print("sample only %.1f us" % timeit(sample))
print("serial      %.1f us" % timeit(serial))
print("overlapped  %.1f us" % timeit(overlapped))
