#!/usr/bin/env python3
"""Mean counter value per (kernel, grid, counter) of a rocprofv3 --pmc counter_collection CSV, as a markdown table.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... --output-format csv -d DIR -o NAME -- <program>
    python tools/pmc_summary.py DIR/NAME_counter_collection.csv [kernel substring] > profiles/rX_pmc.md
"""
import collections
import csv
import re
import sys


def main(path, only=""):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if only and only not in name:
            continue
        name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:60]
        acc[(name, r.get("Grid_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("| kernel | grid | counter | launches | mean |")
    print("|---|---:|---|---:|---:|")
    for (name, grid, ctr), v in sorted(acc.items()):
        print(f"| `{name}` | {grid} | {ctr} | {len(v)} | {sum(v) / len(v):.0f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
