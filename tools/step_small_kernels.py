"""Developer tool: which framework ops launch the SMALL kernels of a bench step (copies, casts, adds that are not the
path's own kernels)?  Runs the step eagerly under torch.profiler and prints, per (aten op, input shapes), how many device
kernels it launched and their device time per step.
    python tools/step_small_kernels.py [--train]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

OWN = ("gemm256", "attention_", "add_layernorm", "bf_sample", "fused_small", "embed_", "param_grad", "gelu_bwd", "colsum",
       "layernorm_param", "bf_reduce", "transpose_kernel")


def main():
    train = "--train" in sys.argv
    step, _, _, bmodel = bench.make_bert(torch.device("cuda"), 10, "bf16", train=train, train_mode=train)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, set()])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        ks = [k for k in ev.kernels if not any(o in k.name for o in OWN)]
        if not ks:
            continue
        parent = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
        key = (ev.name, parent, str(ev.input_shapes)[:90])
        agg[key][0] += len(ks)
        agg[key][1] += sum(k.duration for k in ks)
        agg[key][2].update(k.name[:40] for k in ks)
    tot = 0.0
    for (name, parent, shapes), (n, us, kn) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        tot += us
        print(f"{n:4d} k {us:8.1f} us  {name:22s} < {parent:28s} {shapes:90s} {sorted(kn)[0]}")
    print(f"total {sum(v[1] for v in agg.values()):.1f} us of framework kernels in one step")


if __name__ == "__main__":
    main()
