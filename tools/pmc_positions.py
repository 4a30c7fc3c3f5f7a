#!/usr/bin/env python3
"""Per-position counters of the tiled GEMM launches of a BERT step, from rocprofv3 --pmc passes of bench.py (--graph off):

    python tools/pmc_positions.py <title> <dir with *counter_collection.csv of one or more passes> ... > profiles/rX_pmc_gemm_positions.md

The tiled launches of a step come in the model's order — per encoder layer: Q/K/V, attention-out, FFN-up+GELU, FFN-down
(bench.py GEMM_POSITIONS) — so GEMM dispatch i (in Dispatch_Id order, warm-up steps included: every step has 48) is position
i % 4.  Counters are averaged per launch; SQ_* are summed over the chip's SQs by rocprofv3 (quad-cycles for *_CYCLES and
WAIT_*, cycles for SQ_VALU_MFMA_BUSY_CYCLES — MI355X_MICROARCH.md, "rocprofv3 PMC slots")."""
import collections
import csv
import glob
import os
import sys

POS = ("Q/K/V", "attention-out", "FFN-up+GELU", "FFN-down")
OPERANDS = {"Q/K/V": 10 * (4096 * 768 + 3 * 768 * 768) * 2, "attention-out": 10 * (4096 * 768 + 768 * 768) * 2,
            "FFN-up+GELU": 10 * (4096 * 768 + 3072 * 768) * 2, "FFN-down": 10 * (4096 * 3072 + 768 * 3072) * 2}
FLOP = {"Q/K/V": 2 * 10 * 4096 * 2304 * 768, "attention-out": 2 * 10 * 4096 * 768 * 768,
        "FFN-up+GELU": 2 * 10 * 4096 * 3072 * 768, "FFN-down": 2 * 10 * 4096 * 768 * 3072}


def is_gemm(name):
    return "gemm256_ring5" in name or "gemm256_sched" in name


PATTERN = [0, 1, 2, 3]  # positions of one encoder layer's launches (--pattern 0,1,2,3,2,3: FFN pair in two sample windows)


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if is_gemm(r["Kernel_Name"])]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        rank = {d_: i for i, d_ in enumerate(ids)}
        seen = set()
        for r in rows:
            pos = POS[PATTERN[rank[int(r["Dispatch_Id"])] % len(PATTERN)]]
            acc[pos][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:  # the launch's duration under the counters (kernels run serialised, at their usual length)
                seen.add(r["Dispatch_Id"])
                acc[pos]["duration_us[" + os.path.basename(os.path.normpath(d)) + "]"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc


def main():
    global PATTERN
    argv = sys.argv[1:]
    if "--pattern" in argv:
        i = argv.index("--pattern")
        PATTERN = [int(v) for v in argv[i + 1].split(",")]
        del argv[i:i + 2]
    title, dirs = argv[0], argv[1:]
    acc = collections.defaultdict(dict)
    for d in dirs:
        for pos, ctrs in load(d).items():
            for c, v in ctrs.items():
                acc[pos][c] = (sum(v) / len(v), len(v))
    names = sorted({c for p in acc.values() for c in p})
    print(f"# {title}\n")
    print("| counter (mean per launch) | " + " | ".join(POS) + " |")
    print("|---|" + "---:|" * len(POS))
    for c in names:
        print(f"| {c} | " + " | ".join(f"{acc[p][c][0]:.4g}" if c in acc[p] else "—" for p in POS) + " |")
    print(f"| launches averaged | " + " | ".join(str(acc[p][names[0]][1]) if names and names[0] in acc[p] else "—" for p in POS) + " |")
    g = lambda p, c: acc[p][c][0] if c in acc[p] else None
    print("\n## derived\n")
    print("| | " + " | ".join(POS) + " |")
    print("|---|" + "---:|" * len(POS))
    rows = []

    def row(label, fn):
        vals = []
        for p in POS:
            try:
                v = fn(p)
                vals.append("—" if v is None else (f"{v:.3f}" if abs(v) < 100 else f"{v:.0f}"))
            except (TypeError, ZeroDivisionError):
                vals.append("—")
        rows.append(f"| {label} | " + " | ".join(vals) + " |")

    row("L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS)", lambda p: g(p, "TCC_HIT_sum") / (g(p, "TCC_HIT_sum") + g(p, "TCC_MISS_sum")))
    row("fabric read MB = TCC_EA0_RDREQ x 64 B x 2 (gfx950 correction)", lambda p: g(p, "TCC_EA0_RDREQ_sum") * 128 / 1e6)
    row("... / operand bytes", lambda p: g(p, "TCC_EA0_RDREQ_sum") * 128 / OPERANDS[p])
    # SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_f32_16x16x32 (= flop / 1024: checked), summed over the 1024 SIMDs;
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: SIMD-cycles of the launch = 1024 x GRBM / 8
    row("matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)", lambda p: g(p, "SQ_VALU_MFMA_BUSY_CYCLES") / (128.0 * g(p, "GRBM_GUI_ACTIVE")))
    sqdur = lambda p: next((acc[p][c][0] for c in acc[p] if c.startswith("duration_us[sq")), None)
    row("shader clock GHz = GRBM_GUI_ACTIVE / 8 / duration", lambda p: g(p, "GRBM_GUI_ACTIVE") / 8.0 / sqdur(p) / 1e3)
    row("fraction of the 2.5 PFLOP/s peak = flop / duration", lambda p: FLOP[p] / sqdur(p) / 1e6 / 2500.0)
    row("waves resident = 4 x SQ_WAVE_CYCLES / (2048 waves x GRBM / 8)", lambda p: 4.0 * g(p, "SQ_WAVE_CYCLES") / (256.0 * g(p, "GRBM_GUI_ACTIVE")))
    row("waves parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES", lambda p: g(p, "SQ_WAIT_ANY") / g(p, "SQ_WAVE_CYCLES"))
    row("issue stalls = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES", lambda p: g(p, "SQ_WAIT_INST_ANY") / g(p, "SQ_WAVE_CYCLES"))
    row("... of which LDS = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES", lambda p: g(p, "SQ_WAIT_INST_LDS") / g(p, "SQ_WAVE_CYCLES"))
    row("issuing = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES", lambda p: g(p, "SQ_ACTIVE_INST_ANY") / g(p, "SQ_WAVE_CYCLES"))
    row("kernel clocks = GRBM_GUI_ACTIVE", lambda p: g(p, "GRBM_GUI_ACTIVE"))
    print("\n".join(rows))


if __name__ == "__main__":
    main()
