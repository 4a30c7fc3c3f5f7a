import sys, copy
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import bayeformers_amd as bf
from bayeformers_amd import random as bfr, ops
from bayeformers_amd.training import GraphedTrainingStep, training_step
import test_gpu_backward as T
g, bmodel0, _, inputs, labels = T._tiny_train_setup("/root/repo/tests/golden", "bf16", True)
S, NB = int(g["S"]), int(g["n_batches"])
batches = [inputs, {"input_ids": inputs["input_ids"].flip(0).contiguous(), "attention_mask": inputs["attention_mask"]}]
order = [0, 0, 1, 0, 1]
nll = lambda mean: torch.nn.functional.cross_entropy(mean[0].float(), labels)
bf.set_compute_dtype("bf16")
def fresh():
    m = copy.deepcopy(bmodel0); m.train(False); bf.fuse_attention(m)
    ps = [p for p in m.parameters() if p.requires_grad]
    return m, torch.optim.AdamW(ps, lr=torch.tensor(1e-3, device="cuda"), eps=1e-8, weight_decay=0.0, fused=True, capturable=True)
def run(kind):
    torch.manual_seed(11); bf.manual_seed(T.SEED)
    m, opt = fresh()
    snaps = []
    step = GraphedTrainingStep(m, batches[0], S, nll, opt, NB, max_grad_norm=1.0, eager_steps=2) if kind == "graph" else None
    for b in order:
        ops.COLSUMS_FOLDED[0] = 0
        loss = float(step(batches[b]) if step else training_step(m, batches[b], S, nll, opt, NB, max_grad_norm=1.0))
        snaps.append((loss, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad},
                      {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, ops.COLSUMS_FOLDED[0]))
    if step: step.close()
    return snaps
a, b, c = run("eager"), run("eager"), run("graph")
for name, x, y in (("eager vs eager", a, b), ("eager vs graph", a, c)):
    print(name)
    for k, (sx, sy) in enumerate(zip(x, y)):
        dp = [n for n in sx[1] if not torch.equal(sx[1][n], sy[1][n])]
        dg = [n for n in sx[2] if n in sy[2] and not torch.equal(sx[2][n], sy[2][n])]
        print(f"  step {k}: loss equal {sx[0] == sy[0]}; params differing {len(dp)} {dp[:3]}; grads differing {len(dg)} {dg[:4]}; colsums folded {sx[3]} vs {sy[3]}")
