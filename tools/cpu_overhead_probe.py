"""How much host time does one BERT-base step take to ENQUEUE (vs the 10 ms the GPU needs to run it)?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench

class A: pass
device = torch.device("cuda", 0)
import bayeformers_amd as bf
bf.set_compute_dtype("bf16"); bf.manual_seed(0x5EED)
step, _, cfg, bmodel = bench.make_bert(device, 10, "bf16")
for _ in range(3): step()
torch.cuda.synchronize()
for n in (1, 2, 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueue {1e3*(t1-t0)/n:.2f} ms/step, until done {1e3*(t2-t0)/n:.2f} ms/step")
