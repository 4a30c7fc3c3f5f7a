"""Which host ops launch what during one benchmarked BERT-base step?  torch.profiler over a few steps of
bench.make_bert's step(): per kernel name the launch count per step, and for the copy / elementwise kernels that do not
belong to the path the aten op and the innermost Python frames that issued them.

    python tools/step_trace.py [--steps 3] [--filter copy]
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--filter", default="copy,elementwise,Memcpy,Memset,fill,cat,reduce")
    ap.add_argument("--train", action="store_true", help="the training step (bert_base_train) instead of fwd+ELBO")
    args = ap.parse_args()
    import bayeformers_amd as bf

    dev = torch.device("cuda", 0)
    bf.set_compute_dtype("bf16")
    bf.manual_seed(0x5EED)
    step, _, _, _ = bench.make_bert(dev, 10, "bf16", train=args.train)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    evs = prof.events()
    kernels = collections.Counter()
    by_kernel_src = collections.defaultdict(collections.Counter)
    pats = [p for p in args.filter.split(",") if p]
    # map: a device event carries the correlated CPU op through .cpu_parent / linked launch; use key_averages by stack
    for e in evs:
        if e.device_type == torch.autograd.DeviceType.CUDA:
            kernels[e.name] += 1
    n = args.steps
    print(f"== device activities per step ({sum(kernels.values()) / n:.0f} total)")
    for k, c in kernels.most_common():
        print(f"{c / n:8.1f}  {k[:110]}")
    print("\n== host ops with device time, grouped by innermost stack frames (per step)")
    ka = prof.key_averages(group_by_stack_n=12)
    rows = []
    for a in ka:
        if a.device_time_total <= 0:
            continue
        if not any(p.lower() in a.key.lower() for p in pats):
            continue
        rows.append((a.count / n, a.device_time_total / n, a.key,
                     [s for s in a.stack if "site-packages/torch/" not in s and "dist-packages/torch/" not in s][:4]))
    rows.sort(key=lambda r: -r[0])
    for cnt, us, key, stack in rows[:60]:
        print(f"{cnt:7.1f} x  {us:8.1f} us  {key}")
        for s in stack:
            print("            ", s.strip()[:140])


if __name__ == "__main__":
    main()
