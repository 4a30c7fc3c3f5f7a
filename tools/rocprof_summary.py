#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (…_results.db) as a per-kernel table: calls, total, average, share.

    python tools/rocprof_summary.py gpurun_out/prof_x/bert_results.db > profiles/r1_x.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void at::native::", "at::", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(path, extra=""):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary: {path}")
    if extra:
        print(f"\n{extra}")
    print(f"\ntotal kernel time {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches (durations in us)\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, tot, avg, pct in rows[:24]:
        print(f"| `{short(name)}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
