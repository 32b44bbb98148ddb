#!/usr/bin/env python
"""One row per (kernel, grid) out of rocprofv3 counter-collection CSVs (any number of passes over the same command): counters averaged
over the LAST occurrence block of each kernel, plus derived shares for the SQ pass (parked / issue-stalled / issuing share of the wave
cycles; LDS bank-conflict share of the LDS-active cycles; L2 hit rate for the TCC pass).
    python tools/pmc_kernel_table.py out.md pass1_counter_collection.csv [pass2_counter_collection.csv ...]"""
import collections
import csv
import gzip
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trace_summary as TS  # noqa: E402


def load(path):
    op = gzip.open if path.endswith(".gz") else open
    by = collections.OrderedDict()
    with op(path, "rt") as fh:
        for r in csv.DictReader(fh):
            key = int(r["Dispatch_Id"])
            d = by.setdefault(key, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])),
                                    "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": collections.defaultdict(float)})
            d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    return list(by.values())


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    rows = collections.OrderedDict()
    for p in paths:
        d = load(p)
        d = d[len(d) * 2 // 3:]                                   # the last of the probe's three passes
        for k in d:
            name = TS.gemm_name(k["name"]) or TS.short(k["name"])
            e = rows.setdefault((name, k["grid"]), {"n": collections.defaultdict(int), "c": collections.defaultdict(float), "us": [], })
            e["us"].append(k["us"])
            for cn, v in k["c"].items():
                e["c"][cn] += v
                e["n"][cn] += 1
    lines = ["| kernel | workgroups | launches | avg us (profiled) | parked (s_waitcnt / barrier) | issue-stalled | issuing | LDS-issue stall | MFMA busy | LDS bank-conflict share | L2 hit rate | L2 reads / launch (MB) | fabric reads / launch (MB) |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for (name, grid), e in rows.items():
        c = {k: v / max(1, e["n"][k]) for k, v in e["c"].items()}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)

        def share(k):
            return f"{100.0 * c[k] / wc:.0f} %" if wc and k in c else "-"
        us = sum(e["us"]) / len(e["us"])
        mfma = f"{100.0 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (us * 1e-6 * 2.4e9 * 1024):.0f} %" if "SQ_VALU_MFMA_BUSY_CYCLES" in c else "-"
        bank = f"{100.0 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.0f} %" if c.get("SQ_LDS_IDX_ACTIVE") else "-"
        hit = f"{100.0 * c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.0f} %" if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0 else "-"
        l2 = f"{c['TCP_TCC_READ_REQ_sum'] * 128 / 1e6:.0f}" if "TCP_TCC_READ_REQ_sum" in c else "-"
        ea = f"{c['TCC_EA0_RDREQ_sum'] * 64 / 1e6:.0f}" if "TCC_EA0_RDREQ_sum" in c else "-"
        lines.append(f"| {name} | {grid} | {len(e['us']) // max(1, len(paths))} | {us:.1f} | {share('SQ_WAIT_ANY')} | {share('SQ_WAIT_INST_ANY')} | {share('SQ_ACTIVE_INST_ANY')} | "
                     f"{share('SQ_WAIT_INST_LDS')} | {mfma} | {bank} | {hit} | {l2} | {ea} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
