#!/usr/bin/env python3
"""Per-kernel summary and per-position GEMM breakdown of a rocprofv3 --kernel-trace CSV of bench.py (BERT-base):

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bert -- python3 bench.py --steps 10 ...
    python tools/rocprof_positions.py gpurun_out/prof/**/bert_kernel_trace.csv > profiles/r2x_bert_base.md

The 48 tiled-GEMM launches of a step come in the model's order — per encoder layer: Q/K/V in one launch, attention
output, FFN-up (+GELU), FFN-down — so launch i of a step is position i % 4 of layer i // 4.
"""
import collections
import csv
import re
import sys

FLOP = {"QKV (N=2304, K=768)": 2 * 10 * 4096 * 2304 * 768, "attn-out (N=768, K=768)": 2 * 10 * 4096 * 768 * 768,
        "FFN-up + GELU (N=3072, K=768)": 2 * 10 * 4096 * 3072 * 768, "FFN-down (N=768, K=3072)": 2 * 10 * 4096 * 768 * 3072}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void at::native::", "at::", name)
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    return name if len(name) <= 100 else name[:97] + "..."


def main(path, note=""):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in per.values())
    print(f"# rocprofv3 --kernel-trace summary: {path}\n")
    if note:
        print(note + "\n")
    print(f"total kernel time {total / 1e3:.3f} ms over {len(rows)} dispatches (durations in us)\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:22]:
        print(f"| `{short(name)}` | {len(v)} | {sum(v):.1f} | {sum(v) / len(v):.2f} | {100 * sum(v) / total:.2f} |")
    gemm = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "gemm256_sched" in r["Kernel_Name"] or "gemm256_ring5" in r["Kernel_Name"]]
    if len(gemm) >= 48 and len(gemm) % 48 == 0:
        steps = len(gemm) // 48
        names = list(FLOP)
        pos = collections.defaultdict(list)
        for i, d in enumerate(gemm):
            pos[names[i % 4]].append(d)
        tot_us = sum(gemm) / steps
        tot_flop = 12 * sum(FLOP.values())
        print(f"\n## tiled GEMM by position ({steps} steps x 48 launches)\n")
        print(f"GEMM time per step {tot_us / 1e3:.3f} ms -> {tot_flop / tot_us / 1e6:.0f} TFLOP/s = "
              f"{tot_flop / tot_us / 1e6 / 2500:.4f} of the 2.5 PFLOP/s dense bf16 peak\n")
        print("| position | launches | avg us | min us | TFLOP/s | share of GEMM time |")
        print("|---|---:|---:|---:|---:|---:|")
        for n in names:
            v = pos[n]
            avg = sum(v) / len(v)
            print(f"| {n} | {len(v)} | {avg:.1f} | {min(v):.1f} | {FLOP[n] / avg / 1e6:.0f} | {100 * sum(v) / sum(gemm):.1f} % |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
