import torch, time, sys, os
sys.path.insert(0, os.getcwd())
import bayeformers_amd as bf, bayeformers_amd.nn as bnn
torch.manual_seed(0)
layer = bnn.Linear(768, 768).cuda(); layer.layer_id = 0
x = torch.randn(320, 768, device='cuda').bfloat16()
lp = torch.zeros(10, 2, dtype=torch.float64, device='cuda')
from bayeformers_amd import ops
def run():
    return ops.linear_forward(layer, x, 10, 0x5EED, 0, lp)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("fused small-M linear fwd: %.2f us per call" % (e0.elapsed_time(e1) * 1000 / 50), lp[0].tolist())
